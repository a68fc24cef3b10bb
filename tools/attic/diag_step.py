import sys
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from ha2g_amd import procedural as proc, train_hierarchy as th
from ha2g_amd.config import CASES
from ha2g_amd.optim import FusedAdam
from ha2g_testing import EpsInjector, batch_for, build_modules, named_state
DEV = 'cuda:0'
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg1'
case = CASES[name]
g = np.load('tests/golden/%s.npz' % name)
args, gens, dis, aud, txt = build_modules(case, DEV)
text, spec, target, vid = (t.to(DEV) for t in batch_for(case))
lr = 5e-4
g_opts = [FusedAdam(m.parameters(), lr=lr) for m in gens]
dis_opt = FusedAdam(dis.parameters(), lr=lr * 0.2)
aud_opt, txt_opt = FusedAdam(aud.parameters(), lr=lr), FusedAdam(txt.parameters(), lr=lr)
EpsInjector(gens, case['seed'], case['B'])
perm = torch.from_numpy(proc.fixed_perm(case['B'], case['seed'])).to(DEV)
th.randperm_source = lambda n, device: perm
mods = dict(g1=gens[0], g2=gens[1], g3=gens[2], dis=dis, audio=aud, text=txt)
ret = th.train_iter_hierarchy(args, 0, text, spec, target, vid, *gens, dis, aud, txt, *g_opts, dis_opt, aud_opt, txt_opt)
for k, v in ret.items():
    r = float(g['step0/ret/' + k]); print('ret %-8s %.8f ref %.8f rel %.1e (noise %.1e)' % (k, v, r, abs(v - r) / abs(r), float(g['step0/ret/%s@noise' % k]) / abs(r)))
sd, grads = named_state(mods)
rows = []
for k in g.files:
    if k.startswith('step0/grad/') and k.endswith('/norm') and '.net.' not in k:
        pk = k[11:-5]
        if pk not in grads: continue
        a = grads[pk].detach().double().cpu().numpy().reshape(-1)
        rn = float(g[k]); e = abs(np.sqrt((a * a).sum()) - rn) / max(rn, 1e-30); n = float(g[k + '@noise']) / max(rn, 1e-30)
        rows.append((e, n, pk))
roles = {}
for e, n, pk in rows:
    roles.setdefault(pk.split('.')[0], []).append((e, n))
for r, v in roles.items():
    es, ns = np.array([x[0] for x in v]), np.array([x[1] for x in v])
    print('%-6s n=%3d  err median %.1e max %.1e | ref-noise median %.1e max %.1e' % (r, len(v), np.median(es), es.max(), np.median(ns), ns.max()))
rows.sort(reverse=True)
for e, n, pk in rows[:12]:
    print('  %.1e (noise %.1e) %s' % (e, n, pk))

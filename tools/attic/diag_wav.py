import sys
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from ha2g_amd import procedural as proc
from ha2g_amd.config import CASES
from ha2g_testing import batch_for, build_modules, wproc
DEV = 'cuda:0'
name = sys.argv[1] if len(sys.argv) > 1 else 'small'
g = np.load('tests/golden/%s.npz' % name)
case = CASES[name]
_, _, _, aud, _ = build_modules(case, DEV)
_, spec, _, vid = batch_for(case)
w, lo, mid, hi, blend = aud(spec.to(DEV), vid.to(DEV))
for nm, t in (('weight', w), ('low', lo), ('mid', mid), ('high', hi)):
    ref = g['audio/' + nm]
    print('fwd %-8s rel err %.2e  (ref noise %.2e)' % (nm, np.abs(t.detach().cpu().double().numpy() - ref).max() / np.abs(ref).max(), g['audio/%s@noise' % nm] / np.abs(ref).max()))
s = case['seed']
loss = sum((b * wproc('blend%d' % i, b, s)).sum() for i, b in enumerate(blend)) + (hi * wproc('hi', hi, s)).sum() + (lo * wproc('lo', lo, s)).sum()
loss.backward()
for k, p in aud.named_parameters():
    key = 'audio/grad/' + k
    a = p.grad.detach().cpu().double().numpy().reshape(-1)
    stride = max(1, a.size // 64)
    smp = a[::stride][:64]
    ref = g[key + '/sample']
    rn = float(g[key + '/norm'])
    scale = max(np.abs(ref).max(), rn / max(np.sqrt(a.size), 1.0), 1e-30)
    e_s = np.abs(smp - ref).max() / scale
    n_s = float(g[key + '/sample@noise']) / scale
    e_n = abs(np.sqrt((a * a).sum()) - rn) / rn
    n_n = float(g[key + '/norm@noise']) / rn
    flag = '  <<<' if (e_s > 1e-4 + 4 * n_s or e_n > 1e-4 + 4 * n_n) else ''
    print('%-48s sample err %.1e (noise %.1e)  norm err %.1e (noise %.1e)%s' % (k[15:], e_s, n_s, e_n, n_n, flag))

"""Full-tensor comparison of the generators' embedding gradients (expr_cfg1, step 0) between the HIP step and the CPU oracle in
float64 and float32 run in-process: where does the error sit (row 0 = padding id vs word rows), how does the oracle's own fp32 compare."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from ha2g_amd import procedural as proc, schema, train_hierarchy as th
from ha2g_amd.config import CASES, EXPRESSIVE_SPEC, make_args
from ha2g_amd.optim import FusedAdam
from ha2g_testing import EpsInjector, batch_for, build_modules, state_for
from oracle import ha2g_oracle as O

name = sys.argv[1] if len(sys.argv) > 1 else 'expr_cfg1'
case = CASES[name]
expr = bool(case.get('expressive'))
dims = schema.EXPRESSIVE_POSE_DIMS if expr else schema.GESTURE_POSE_DIMS
P = dims[-1]
DEV = 'cuda:0'
args, gens, dis, aud, txt = build_modules(case, DEV, dims)
import os
OFF = os.environ.get('HA2G_DIAG_OFF', '').split(',')
WMAP = dict(phys='loss_physical_weight', cpos='loss_contrastive_pos_weight', cneg='loss_contrastive_neg_weight', div='loss_reg_weight', kld='loss_kld_weight')
for o in OFF:
    if o in WMAP:
        setattr(args, WMAP[o], 0.0)
print('weights off:', OFF)
text, spec, target, vid = (t.to(DEV) for t in batch_for(case, P=P))
lr = float(args.learning_rate)
g_opts = [FusedAdam(m.parameters(), lr=lr) for m in gens]
dis_opt = FusedAdam(dis.parameters(), lr=lr * args.discriminator_lr_weight)
aud_opt, txt_opt = FusedAdam(aud.parameters(), lr=lr), FusedAdam(txt.parameters(), lr=lr)
EpsInjector(gens, case['seed'], case['B'])
perm = torch.from_numpy(proc.fixed_perm(case['B'], case['seed'])).to(DEV)
th.randperm_source = lambda n, device: perm
fn = th.train_iter_hierarchy_expressive if expr else th.train_iter_hierarchy
fn(args, 0, text, spec, target, vid, *gens, dis, aud, txt, *g_opts, dis_opt, aud_opt, txt_opt)
got = {('g%d' % (i + 1)): g.text_encoder.embedding.weight.grad.detach().double().cpu() for i, g in enumerate(gens)}
got['text'] = txt.embedding.weight.grad.detach().double().cpu()
refs = {}
for dt in (torch.float64, torch.float32):
    sd = state_for(case, dt, dims)
    oargs = make_args(case)
    for o in OFF:
        if o in WMAP:
            setattr(oargs, WMAP[o], 0.0)
    tr = O.OracleTrainer(sd, oargs, EXPRESSIVE_SPEC if expr else None)
    es = proc.EpsStream(case['seed'])
    t2, s2, g2, v2 = batch_for(case, dt, P=P)
    tr.train_iter(0, t2, s2, g2, v2, lambda shp: torch.from_numpy(es(shp)).to(dt), torch.from_numpy(proc.fixed_perm(case['B'], case['seed'])))
    refs[dt] = {k: tr.grads[k + '.text_encoder.embedding.weight' if k != 'text' else 'text.embedding.weight'].double() for k in got}
for k in got:
    r64, r32 = refs[torch.float64][k], refs[torch.float32][k]
    sc = float(r64.abs().max())
    e_hip, e_o32 = (got[k] - r64).abs(), (r32 - r64).abs()
    i = int(e_hip.argmax()); row, col = divmod(i, r64.shape[1])
    print('%-5s scale %.3e | HIP max %.2e (row %d col %d, truth %.3e) row0 %.2e rest %.2e | oracle-fp32 max %.2e row0 %.2e rest %.2e' % (
        k, sc, float(e_hip.max()), row, col, float(r64[row, col]), float(e_hip[0].max()), float(e_hip[1:].max()),
        float(e_o32.max()), float(e_o32[0].max()), float(e_o32[1:].max())))

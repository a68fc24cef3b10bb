"""Stand-alone generator forward/backward (HIP) vs the oracle in float64 for several pose widths: which tensors carry error."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from ha2g_amd import procedural as proc, schema
from ha2g_amd.config import CASES
from ha2g_testing import build_modules, batch_for, state_for, wproc, leaf_params
from oracle import ha2g_oracle as O

DEV = 'cuda:0'
case = CASES['expr_cfg1']
dims = schema.EXPRESSIVE_POSE_DIMS
args, gens, dis, aud, txt = build_modules(case, DEV, dims)
B = case['B']
text, spec, target, vid = batch_for(case, P=126)
for k in (int(a) for a in sys.argv[1:]) if len(sys.argv) > 1 else range(6):
    P = dims[k]
    role = 'g%d' % (k + 1)
    g = gens[k]
    eps = torch.from_numpy(proc.EpsStream(case['seed'])((B, 16)))
    g.eps_source = lambda shape, device: eps.to(device)
    pre0 = torch.from_numpy(proc.tensor_for('in.pre%d' % k, (B, 34, P + 1), 3) * 5)
    af0 = torch.from_numpy(proc.tensor_for('in.afeat', (B, 34, 32), case['seed']) * 10)
    pre, af = pre0.to(DEV).requires_grad_(True), af0.to(DEV).requires_grad_(True)
    for p_ in g.parameters():
        p_.grad = None
    o, z, mu, lv = g(pre, text.to(DEV), af, vid.to(DEV))
    w = wproc('gen', o, case['seed'])
    (o * w).sum().backward()
    sd = state_for(case, torch.float64, dims)
    ps = leaf_params(sd, role)
    pre64, af64 = pre0.double().requires_grad_(True), af0.double().requires_grad_(True)
    o64, *_ = O.pose_generator(pre64, text, af64, vid, sd, role + '.', case['n_layers'], case['hidden_size'], eps.double())
    gr = torch.autograd.grad((o64 * w.double().cpu()).sum(), [pre64, af64] + list(ps.values()))
    def rel(a, b):
        return float((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30))
    rows = [('out', rel(o.detach(), o64.detach())), ('grad_pre', rel(pre.grad, gr[0])), ('grad_afeat', rel(af.grad, gr[1]))]
    for (name, _), g64 in zip(ps.items(), gr[2:]):
        p_ = dict(g.named_parameters())[name[len(role) + 1:]]
        rows.append((name[len(role) + 1:], rel(p_.grad, g64)))
    rows.sort(key=lambda r: -r[1])
    print('P=%d in_size=%d :' % (P, P + 81), ' '.join('%s=%.1e' % r for r in rows[:7]))

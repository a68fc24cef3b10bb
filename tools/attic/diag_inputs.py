import sys
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from ha2g_amd import ops, procedural as proc
from ha2g_amd.config import CASES
from ha2g_testing import batch_for, build_modules, wproc
DEV = 'cuda:0'
for name in ('small', 'cfg1'):
    case = CASES[name]; g = np.load('tests/golden/%s.npz' % name)
    N = case['B'] * 34
    for expr, tag in ((False, 'contrastive'), (True, 'contrastive_expr')):
        a = torch.from_numpy(proc.tensor_for('in.ca', (N, 32), case['seed']) * 6).to(DEV).requires_grad_(True)
        b = torch.from_numpy(proc.tensor_for('in.cb', (N, 32), case['seed']) * 6).to(DEV).requires_grad_(True)
        l = ops.contrastive(a, b, expr); l.backward()
        for nm, t in (('loss', l), ('grad_a', a.grad), ('grad_b', b.grad)):
            ref = g['%s/%s' % (tag, nm)]
            print(name, tag, nm, 'rel err %.2e  ref-noise %.2e' % (np.abs(t.detach().cpu().double().numpy() - ref).max() / np.abs(ref).max(), float(g['%s/%s@noise' % (tag, nm)]) / np.abs(ref).max()))
    _, gens, _, _, _ = build_modules(case, DEV)
    g3 = gens[2]
    text, _, target, vid = batch_for(case)
    B = case['B']
    eps = torch.from_numpy(proc.EpsStream(case['seed'])((B, 16))).to(DEV)
    g3.eps_source = lambda shape, device: eps
    pre = torch.zeros(B, 34, 28); pre[:, :4, :-1] = target[:, :4]; pre[:, :4, -1] = 1
    pre = pre.to(DEV).requires_grad_(True)
    afeat = torch.from_numpy(proc.tensor_for('in.afeat', (B, 34, 32), case['seed']) * 10).to(DEV).requires_grad_(True)
    o, z, mu, lv = g3(pre, text.to(DEV), afeat, vid.to(DEV))
    s = case['seed']
    ((o * wproc('gen', o, s)).sum() + (z * wproc('z', z, s)).sum() + (mu * lv).sum()).backward()
    for nm, t in (('out', o), ('grad_pre', pre.grad), ('grad_afeat', afeat.grad)):
        ref = g['gen/' + nm]
        print(name, 'gen', nm, 'rel err %.2e  ref-noise %.2e' % (np.abs(t.detach().cpu().double().numpy() - ref).max() / np.abs(ref).max(), float(g['gen/%s@noise' % nm]) / np.abs(ref).max()))

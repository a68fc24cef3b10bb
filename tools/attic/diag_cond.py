"""Conditioning probe: how much do the audio-encoder gradients of one train step move when the input spectrogram
is perturbed by one float32 ulp-scale relative noise?  (fp32 rounding acts like such a perturbation at every layer.)"""
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


def run(perturb):
    args, gens, dis, aud, txt = build_modules(case, DEV)
    text, spec, target, vid = (t.to(DEV) for t in batch_for(case))
    if perturb:
        r = torch.from_numpy(np.random.Generator(np.random.PCG64(99)).standard_normal(tuple(spec.shape)).astype(np.float32)).to(DEV)
        spec = spec * (1 + perturb * r)
    lr = 5e-4
    g_opts = [FusedAdam(m.parameters(), lr=lr) for m in gens]
    dis_opt = FusedAdam(dis.parameters(), lr=lr * 0.2)
    aud_opt, txt_opt = FusedAdam(aud.parameters(), lr=lr), FusedAdam(txt.parameters(), lr=lr)
    EpsInjector(gens, case['seed'], case['B'])
    perm = torch.from_numpy(proc.fixed_perm(case['B'], case['seed'])).to(DEV)
    th.randperm_source = lambda n, device: perm
    th.train_iter_hierarchy(args, 0, text, spec, target, vid, *gens, dis, aud, txt, *g_opts, dis_opt, aud_opt, txt_opt)
    _, grads = named_state(dict(audio=aud, g3=gens[2]))
    return {k: v.detach().double().cpu() for k, v in grads.items()}


a, b, c = run(0), run(0), run(6e-8)
for role in ('audio', 'g3'):
    det = [float((a[k] - b[k]).abs().max()) for k in a if k.startswith(role)]
    rel = [float((a[k].norm() - c[k].norm()).abs() / a[k].norm().clamp_min(1e-30)) for k in a if k.startswith(role)]
    print('%s %s: run-to-run max abs diff %.1e | norm change under 6e-8 input perturbation: median %.1e max %.1e' % (
        name, role, max(det), np.median(rel), max(rel)))

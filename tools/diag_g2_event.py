"""Where do two exact-fp32 runs of the cfg3_b128 step that differ by one-ulp input moves part ways inside generator 2?  (tools/diag_step_margins.py found a
binary event: 4 of 12 draws put every gradient of g2 8.6e-4 of its scale away from the other 8.)  Forward hooks on g2's sub-modules + the Huber call."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
from ha2g_amd import ops, procedural as proc, schema, train_hierarchy as th
from ha2g_amd._lib import DEFAULT_GEMM_MODE, lib
from ha2g_amd.config import BIG_CASES
from ha2g_amd.optim import FusedAdam
from ha2g_testing import EpsInjector, batch_for, build_modules

DEV = 'cuda:0'
case = BIG_CASES['cfg3_b128']
dims = schema.EXPRESSIVE_POSE_DIMS
GI = int(os.environ.get('DIAG_GEN', 1))


def run(draw, mode=0):
    lib.ha2g_gemm_set_mode(mode)
    rec = {}
    args, gens, dis, aud, txt = build_modules(case, DEV, dims)
    text, spec, target, vid = (t.to(DEV) for t in batch_for(case, P=dims[-1]))
    gen = torch.Generator(device=DEV); gen.manual_seed(1000 + draw)
    for t in (spec, target):
        t.view(torch.int32).add_(torch.randint(-1, 2, t.shape, generator=gen, device=DEV, dtype=torch.int32))
    lr = float(args.learning_rate)
    g_opts = [FusedAdam(m.parameters(), lr=lr) for m in gens]
    dis_opt = FusedAdam(dis.parameters(), lr=lr * args.discriminator_lr_weight)
    aud_opt, txt_opt = FusedAdam(aud.parameters(), lr=lr), FusedAdam(txt.parameters(), lr=lr)
    EpsInjector(gens, case['seed'], case['B'])
    perm = torch.from_numpy(proc.fixed_perm(case['B'], case['seed'])).to(DEV)
    g = gens[GI]
    cnt = {}
    def hook(name):
        def f(mod, inp, out):
            o = out[0] if isinstance(out, tuple) else out
            n = cnt[name] = cnt.get(name, 0) + 1
            if torch.is_tensor(o):
                rec['%s#%d' % (name, n)] = o.detach().double().cpu()
                if o.requires_grad:
                    o.register_hook(lambda gr, key='%s#%d.GRAD' % (name, n): rec.__setitem__(key, gr.detach().double().cpu()))
                for j, t in enumerate(inp):
                    if torch.is_tensor(t) and t.dtype.is_floating_point:
                        rec['%s#%d.in%d' % (name, n, j)] = t.detach().double().cpu()
        return f
    for name, m in g.named_modules():
        if name:
            m.register_forward_hook(hook(name))
    g.register_forward_hook(hook('GEN'))
    old_h = ops.huber
    hc = [0]
    def huber(x, y, beta):
        rec['huber%d.x' % hc[0]] = x.detach().double().cpu(); rec['huber%d.y' % hc[0]] = y.detach().double().cpu()
        hc[0] += 1
        return old_h(x, y, beta)
    ops.huber = huber
    oldp = th.randperm_source
    th.randperm_source = lambda n, device: perm
    try:
        th.train_iter_hierarchy_expressive(args, 0, text, spec, target, vid, *gens, dis, aud, txt, *g_opts, dis_opt, aud_opt, txt_opt)
    finally:
        th.randperm_source = oldp; ops.huber = old_h
        lib.ha2g_gemm_set_mode(DEFAULT_GEMM_MODE)
    for k, p in g.named_parameters():
        if p.grad is not None:
            rec['grad:' + k] = p.grad.detach().double().cpu()
    return rec


da, db = int(os.environ.get('DIAG_A', 0)), int(os.environ.get('DIAG_B', 4))
A, B = run(da), run(db)
print('draw %d vs draw %d (exact fp32 mode); relative = max|a-b| / max|a|' % (da, db))
for k in A:
    if k in B and A[k].shape == B[k].shape:
        d = (A[k] - B[k]).abs()
        s = float(A[k].abs().max().clamp_min(1e-30))
        rel = float(d.max()) / s
        flag = ' <<<' if rel > 2e-5 else ''
        where = ''
        if k.startswith('grad:text') or k.startswith('grad:gru'):
            continue
        if rel > 2e-5 and d.dim() >= 2:
            idx = np.unravel_index(int(d.argmax()), d.shape)
            rows = (d.reshape(d.shape[0], -1).max(1).values > 0.1 * d.max()).nonzero().flatten().tolist()
            where = ' at %s; leading-dim rows within 10x of the max: %s' % (idx, rows[:12])
        print('%-44s %-22s rel %.3e%s%s' % (k, tuple(A[k].shape), rel, flag, where))

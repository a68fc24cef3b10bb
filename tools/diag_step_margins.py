"""Which arithmetic switch moves which gradient digest: step 0 of a full-size fixture under several configurations, the largest error / tolerance
margins of each (Checker's own tolerance).  Run on the GPU box: python tools/diag_step_margins.py cfg3_b128 [key-substring]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import torch
from ha2g_amd import ops, procedural as proc, schema, train_hierarchy as th, wav_engine
from ha2g_amd._lib import DEFAULT_GEMM_MODE, lib
from ha2g_amd.config import BIG_CASES
from ha2g_amd.optim import FusedAdam
from ha2g_testing import Checker, EpsInjector, batch_for, build_modules, named_state

DEV = 'cuda:0'
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg3_b128'
sub = sys.argv[2] if len(sys.argv) > 2 else 'speaker_embedding.1.weight'
case = BIG_CASES[name]
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', name + '.npz'))
expressive = bool(case.get('expressive'))
dims = schema.EXPRESSIVE_POSE_DIMS if expressive else schema.GESTURE_POSE_DIMS


def run(mode, draw=None, **sw):
    old = {}
    for k, v in sw.items():
        m = wav_engine if hasattr(wav_engine, k) else ops
        old[k] = (m, getattr(m, k)); setattr(m, k, v)
    lib.ha2g_gemm_set_mode(mode)
    try:
        ck = Checker(g)
        args, gens, dis, aud, txt = build_modules(case, DEV, dims)
        text, spec, target, vid = (t.to(DEV) for t in batch_for(case, P=dims[-1]))
        if draw is not None:                                # one-ulp moves of the float inputs, as the reference's noise runs make
            gen = torch.Generator(device=DEV); gen.manual_seed(1000 + draw)
            for t in (spec, target):
                t.view(torch.int32).add_(torch.randint(-1, 2, t.shape, generator=gen, device=DEV, dtype=torch.int32))
        lr = float(args.learning_rate)
        g_opts = [FusedAdam(m.parameters(), lr=lr) for m in gens]
        dis_opt = FusedAdam(dis.parameters(), lr=lr * args.discriminator_lr_weight)
        aud_opt, txt_opt = FusedAdam(aud.parameters(), lr=lr), FusedAdam(txt.parameters(), lr=lr)
        EpsInjector(gens, case['seed'], case['B'])
        perm = torch.from_numpy(proc.fixed_perm(case['B'], case['seed'])).to(DEV)
        mods = {'g%d' % (i + 1): m for i, m in enumerate(gens)}
        mods.update(dis=dis, audio=aud, text=txt)
        fn = th.train_iter_hierarchy_expressive if expressive else th.train_iter_hierarchy
        oldp = th.randperm_source
        th.randperm_source = lambda n, device: perm
        try:
            fn(args, 0, text, spec, target, vid, *gens, dis, aud, txt, *g_opts, dis_opt, aud_opt, txt_opt)
        finally:
            th.randperm_source = oldp
        sd, grads = named_state(mods)
        Checker.margins.clear()
        for k, v in grads.items():
            if k.startswith('dis.'):
                continue
            try:
                ck.digest(v, 'step0/grad/' + k)
            except AssertionError:
                pass
        m = sorted([(a if a == a else 1e9, b) for a, b in Checker.margins], reverse=True)
        return m
    finally:
        lib.ha2g_gemm_set_mode(DEFAULT_GEMM_MODE)
        for k, (mod, v) in old.items():
            setattr(mod, k, v)


configs = [('mode 0', 0, {}), ('mode 6', 6, {}), ('mode 70', 70, {}), ('70, GRU_FWD3 off', 70, dict(GRU_FWD3=False)),
           ('70, PLANE_GEMM off', 70, dict(PLANE_GEMM=False)), ('70, GRU_MERGE_DIRS off', 70, dict(GRU_MERGE_DIRS=False)),
           ('70, FWD3 (tower) off', 70, dict(FWD3=False)),
           ('70, all three off', 70, dict(GRU_FWD3=False, PLANE_GEMM=False, GRU_MERGE_DIRS=False))]
if os.environ.get('DIAG_NAN'):                            # every fresh allocation poisoned: an uninitialised read turns the digests NaN
    _empty = torch.empty
    def poisoned(*a, **k):
        t = _empty(*a, **k)
        if t.is_cuda:
            if t.dtype.is_floating_point:
                t.fill_(float('nan'))
            elif t.dtype == torch.uint8:
                t.fill_(255)
        return t
    torch.empty = poisoned
if os.environ.get('DIAG_DRAWS'):                          # scatter of our own arithmetic under one-ulp input moves
    for mode in (0, 70):
        for d in range(int(os.environ['DIAG_DRAWS'])):
            m = run(mode, draw=d)
            print('mode %d draw %d: %s' % (mode, d, '  '.join('%.2f %s' % (a, b.replace('step0/grad/', '')) for a, b in m[:3])), flush=True)
    sys.exit(0)
if os.environ.get('DIAG_ONLY'):
    configs = [configs[int(i)] for i in os.environ['DIAG_ONLY'].split(',')]
for label, mode, sw in configs:
    m = run(mode, **sw)
    print('%-26s top: %s' % (label, '  '.join('%.2f %s' % (a, b.replace('step0/grad/', '')) for a, b in m[:4])))
    print('%-26s  sel: %s' % ('', '  '.join('%.2f %s' % (a, b.replace('step0/grad/', '')) for a, b in m if sub in b and b.endswith('/sample'))))

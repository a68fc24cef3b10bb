#!/usr/bin/env python3
"""More one-ulp-perturbed float32 runs of the REFERENCE for the headline-size step fixtures (cfg2_b128 / cfg3_b128): the `@noise` entries of those
fixtures were the maximum over FIVE reference runs (the plain one + 4 perturbed), and the tail study of round 3 (gen_tail_study.py) showed
that the reference's own fp32 scatter on this network is heavy-tailed -- a maximum over five draws under-estimates it.  This script adds NEXTRA
perturbed float32 runs (new draws, same probe as gen_golden.main_big: inputs and embedding tables moved by one float32 ulp), computes their
deviation from the STORED float64 truth and raises `@noise` where a new run deviates more.  The truth and `@cond` are untouched; `@noise` can
only grow, and only by what the reference itself does.  Container-only (imports /root/reference through gen_golden.py).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_more_noise.py cfg3_b128 [NEXTRA=8] [first draw=5]
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_more_noise.py enc16 60 9        (the B=16 whole-encoder fixture: 9 runs so far)
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_more_noise.py cfg3_b128_gan 32 9 (round 5: the GAN-phase-first fixtures start with 9 runs)
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402
from ha2g_amd.config import BIG_CASES  # noqa: E402

name = sys.argv[1]
nextra = int(sys.argv[2]) if len(sys.argv) > 2 else 8
first = int(sys.argv[3]) if len(sys.argv) > 3 else 5
gan = name.endswith('_gan')                              # the GAN-phase-first fixtures (gen_golden.main_gan): one step at epoch 11 from fresh state
case = BIG_CASES.get(name[:-4] if gan else name)         # None: one of gen_golden.EXTRA's well-conditioned fixtures (enc16, blocks, ...)
path = os.path.join(HERE, name + '.npz')
fix = dict(np.load(path))
grown = 0
for i in range(nextra):
    G.PERTURB_DRAW = first + i
    o = {}
    t0 = time.time()
    if case is not None:
        G.step_goldens(case, o, torch.float32, expressive=bool(case.get('expressive')), perturb=6e-8, perturb_text=True, epochs=(11,) if gan else (0, 11))
    else:
        for fn in G.EXTRA[name]:
            fn(o, torch.float32, perturb=6e-8)
    n = 0
    for k, v in o.items():
        if k not in fix or k + '@noise' not in fix:
            continue
        d = float(np.abs(np.asarray(v, np.float64) - fix[k]).max())
        if d > float(fix[k + '@noise']):
            fix[k + '@noise'] = np.float64(d)
            n += 1
    grown += n
    print('draw %d: %.0f s, %d of %d noise floors raised' % (first + i, time.time() - t0, n, len(o)), flush=True)
    np.savez_compressed(path, **fix)                      # after every run: an interrupted study keeps what it has measured
fix['noise_runs'] = np.float64(float(fix.get('noise_runs', (9 if gan else 5) if case is not None else 9)) + nextra)
np.savez_compressed(path, **fix)
print('wrote', path, '(%d floors raised in total)' % grown)

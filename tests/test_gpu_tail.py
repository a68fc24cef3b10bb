"""Forward-arithmetic variants of the audio tower judged against the REFERENCE's own float32 scatter (round 3).

tests/golden/cfg1_tail.npz holds, for every audio-encoder gradient digest of step 0 of the chaotic B=4 `cfg1` case, the deviations from the
float64 truth of 200 float32 runs of the reference (train_eval/train_hierarchy.py:71-293 through model/ResNetSE34V2.py:118-218), each with its
inputs perturbed by one ulp (tests/golden/gen_tail_study.py).  Round 2 froze the tower's forward summation order because the equally accurate
3-piece forward split (mode 14) put ONE tensor (layer4.0.downsample.0.weight) at 1.04x a tolerance built from the maximum over 25 runs of this
heavy-tailed quantity.  With 200 runs the tail is measured: that tensor's reference deviations reach 8.4e-4 (max of 25: 1.5e-4; q99: 7.8e-4).
Criterion here, for every tensor:   |x - truth64| <= 1e-4 * scale + 1.5 * max over the 200 reference runs.
Default mode 6 and exact-fp32 mode 0 sit at 0.67 of it, mode 14 at 0.84: every variant is inside the reference's own scatter.  (Mode 14 stays
opt-in for a different reason: since the direct 32-channel kernel became bit-identical and default it no longer buys time -- 42.98 vs 42.77 ms.)
"""
import numpy as np
import pytest
import torch

from ha2g_amd import procedural as proc
from ha2g_amd import train_hierarchy as th
from ha2g_amd._lib import DEFAULT_GEMM_MODE, lib
from ha2g_amd.config import CASES
from ha2g_amd.optim import FusedAdam
from ha2g_testing import EpsInjector, batch_for, build_modules

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('mode', [70, 6, 0, 14])
def test_audio_gradients_within_the_reference_tail(golden, mode):
    truth, tail = golden('cfg1'), golden('cfg1_tail')
    assert int(tail['n_runs']) >= 200
    case = CASES['cfg1']
    lib.ha2g_gemm_set_mode(mode)
    args, gens, dis, aud, txt = build_modules(case, DEV)
    text, spec, target, vid = (t.to(DEV) for t in batch_for(case))
    lr = float(args.learning_rate)
    g_opts = [FusedAdam(m.parameters(), lr=lr) for m in gens]
    dis_opt = FusedAdam(dis.parameters(), lr=lr * args.discriminator_lr_weight)
    aud_opt, txt_opt = FusedAdam(aud.parameters(), lr=lr), FusedAdam(txt.parameters(), lr=lr)
    EpsInjector(gens, case['seed'], case['B'])
    perm = torch.from_numpy(proc.fixed_perm(case['B'], case['seed'])).to(DEV)
    old = th.randperm_source
    th.randperm_source = lambda n, device: perm
    try:
        th.train_iter_hierarchy(args, 0, text, spec, target, vid, *gens, dis, aud, txt, *g_opts, dis_opt, aud_opt, txt_opt)
    finally:
        th.randperm_source = old
        lib.ha2g_gemm_set_mode(DEFAULT_GEMM_MODE)
    worst, n = (0.0, None), 0
    for k, p in aud.named_parameters():
        key = 'step0/grad/audio.%s' % k
        if key + '/sample@dev' not in tail.files:
            continue
        a = p.grad.detach().double().cpu().contiguous().numpy().reshape(-1)
        got = {'/norm': np.sqrt((a * a).sum()), '/sample': a[::max(1, a.size // 64)][:64]}
        for suffix, v in got.items():
            ref = truth[key + suffix]
            scale = max(float(np.abs(ref).max()), 1e-30)
            tol = 1e-4 * scale + 1.5 * float(tail[key + suffix + '@dev'].max())
            err = float(np.abs(v - ref).max())
            worst = max(worst, (err / tol, key + suffix))
            n += 1
    assert n >= 390
    print('mode %d: worst error / (1e-4 scale + 1.5 max of 200 reference runs) = %.3f  (%s)' % (mode, worst[0], worst[1]))
    assert worst[0] <= 1.0, worst

"""Sliding-window hierarchical synthesis (scripts/synthesize_hierarchy.py:36-215) against tests/golden/synth.npz, the
output of the reference's own generate_gestures_hierarchy on a 9 s synthetic clip (5 windows, 4-frame cross-faded overlaps).
The reference function casts to float32 itself, so the fixture is its float32 result with the scatter of eight one-ulp-perturbed runs."""
import numpy as np
import pytest
import torch

from ha2g_amd import procedural as proc
from ha2g_amd.config import CASES, GESTURE_SPEC, SYNTH_CASE, make_args
from ha2g_testing import state_for


class SynthLang:
    SOS_token, EOS_token = 1, 2

    def get_word_index(self, word):
        return 4 + int(word[1:])


def _inputs():
    sc = SYNTH_CASE
    n_audio = int(sc['clip_seconds'] * 16000)
    return sc, n_audio, proc.synth_spectrogram(n_audio, sc['seed']), proc.synth_words(sc['clip_seconds'], sc['n_words'], sc['seed'])


def _check(got, g, key='synth/out'):
    ref = g[key]
    assert got.shape == ref.shape and ref.shape[0] == 154
    tol = 1e-4 * np.abs(ref).max() + 3 * float(g[key + '@noise'])
    err = np.abs(np.asarray(got, np.float64) - ref).max()
    assert err <= tol, (err, tol)


def test_window_plan_and_tokens():
    from ha2g_amd.synthesize import calc_spectrogram_length_from_motion_length, frame_tokens, plan_windows
    assert calc_spectrogram_length_from_motion_length(34, 15) == 70
    unit, stride, n = plan_windows(9.0, 34, 4, 15)
    assert (round(unit, 4), round(stride, 4), n) == (2.2667, 2.0, 5)
    assert plan_windows(1.0, 34, 4, 15)[2] == 1
    words = [('w3', 0.1, 0.3), ('w7', 1.0, 1.2), ('w9', 2.5, 2.9)]
    tok = frame_tokens(SynthLang(), words, 0.0, 34 / 15, 34)[0]
    assert tok[1] == 7 and tok[15] == 11 and int((tok != 0).sum()) == 2          # w9 starts after the window


@pytest.mark.parametrize('expressive', [False, True])
def test_oracle_synthesis_matches_reference(golden, expressive):
    from ha2g_amd import schema
    from ha2g_amd.config import EXPRESSIVE_SPEC
    from oracle import ha2g_oracle as O
    sc, n_audio, spectro, words = _inputs()
    case = CASES['expr_small' if expressive else 'small']
    es = proc.EpsStream(sc['seed'])
    sd = state_for(case, dims=schema.EXPRESSIVE_POSE_DIMS) if expressive else state_for(case)
    out = O.synthesize_windows(make_args(case), sd, EXPRESSIVE_SPEC if expressive else GESTURE_SPEC, SynthLang(), n_audio, words,
                               torch.from_numpy(spectro), sc['vid'], lambda shp: torch.from_numpy(es(shp)))
    _check(out, golden('synth'), 'synth_expr/out' if expressive else 'synth/out')


@pytest.mark.gpu
@pytest.mark.parametrize('expressive', [False, True])
def test_gpu_synthesis_matches_reference(golden, expressive):
    from ha2g_amd import schema
    from ha2g_amd.synthesize import generate_gestures_hierarchy
    from ha2g_testing import build_modules
    sc, n_audio, spectro, words = _inputs()
    case = CASES['expr_small' if expressive else 'small']
    args, gens, dis, aud, txt = build_modules(case, 'cuda:0', schema.EXPRESSIVE_POSE_DIMS) if expressive else build_modules(case, 'cuda:0')
    es = proc.EpsStream(sc['seed'])
    for g_ in gens:                              # reparameterisation noise in the reference's call order g1, g2, g3 per window
        g_.eps_source = lambda shape, device: torch.from_numpy(es(shape)).to(device)
    out = generate_gestures_hierarchy(args, gens, aud, SynthLang(), np.zeros(n_audio, np.float32), words, vid=sc['vid'],
                                      spectrogram=spectro)
    _check(out, golden('synth'), 'synth_expr/out' if expressive else 'synth/out')


@pytest.mark.gpu
def test_window_blend_kernel_is_the_reference_arithmetic():
    from ha2g_amd.synthesize import window_blend
    r = np.random.Generator(np.random.PCG64(3))
    wins = [r.standard_normal((34, 27)).astype(np.float32) for _ in range(4)]
    out_list = []
    for w in wins:                               # synthesize_hierarchy.py:150-161, numpy float32
        seq = w.copy()
        if out_list:
            last = out_list[-1][-4:]
            out_list[-1] = out_list[-1][:-4]
            for j in range(4):
                seq[j] = last[j] * (4 - j) / 5 + seq[j] * (j + 1) / 5
        out_list.append(seq)
    ref = np.vstack(out_list)
    out = torch.zeros(3 * 30 + 34, 27, device='cuda:0')
    for i, w in enumerate(wins):
        window_blend(torch.from_numpy(w).cuda(), out, i, 4)
    assert np.array_equal(out.cpu().numpy(), ref)

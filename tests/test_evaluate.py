"""Validation loop (scripts/train.py:326-500 evaluate_testset, hierarchy branch) against tests/golden/evalset.npz -- the
dict the reference's own function returned for two synthetic loader batches of the `small` case with the FGD evaluator attached."""
import numpy as np
import pytest
import torch

from ha2g_amd import procedural as proc
from ha2g_amd.config import CASES, EVAL_CASE, FGD_CASE, hierarchy_args
from ha2g_amd.evaluate import convert_dir_vec_to_pose, dir_vec_pairs


def test_dir_vec_to_pose_walks_the_bone_chain():
    r = np.random.Generator(np.random.PCG64(5))
    vec = r.standard_normal((2, 7, 27))
    pose = convert_dir_vec_to_pose(vec)
    assert pose.shape == (2, 7, 10, 3) and not pose[..., 0, :].any()
    v = vec.reshape(2, 7, 9, 3)
    for j, (a, b, ln) in enumerate(dir_vec_pairs):                      # every bone is its direction vector scaled by the bone length
        assert np.allclose(pose[..., b, :] - pose[..., a, :], ln * v[..., j, :], rtol=0, atol=1e-15)
    assert np.array_equal(convert_dir_vec_to_pose(v), pose)             # (…, 9, 3) input form
    assert np.array_equal(convert_dir_vec_to_pose(vec[0, 0]), pose[0, 0])


def test_fixture_is_a_full_return_dict(golden):
    g = golden('evalset')
    assert all(t + k in g for t in ('evalset/', 'evalset_expr/') for k in ('loss', 'joint_mae', 'frechet', 'feat_dist', 'diversity'))


@pytest.mark.gpu
@pytest.mark.parametrize('expressive', [False, True])
def test_gpu_evaluate_testset_matches_reference(golden, expressive):
    from ha2g_amd.embedding_space_evaluator import EmbeddingSpaceEvaluator
    from ha2g_amd.evaluate import evaluate_testset
    from ha2g_testing import build_modules
    from ha2g_testing import fgd_ae_state as ae_state
    from ha2g_amd import schema
    g, ec, case = golden('evalset'), EVAL_CASE, CASES['expr_small' if expressive else 'small']
    tag, P = ('evalset_expr', 126) if expressive else ('evalset', 27)
    args, gens, dis, aud, txt = build_modules(case, 'cuda:0', schema.EXPRESSIVE_POSE_DIMS) if expressive else build_modules(case, 'cuda:0')
    ckpt = {'pose_dim': 126, 'latent_dim': 128, 'motion_ae': ae_state(P=126)} if expressive else {'pose_dim': 27, 'gen_dict': ae_state()}
    ev = EmbeddingSpaceEvaluator(hierarchy_args(expressive=expressive), ckpt, None, 'cuda:0')
    batches = []
    for i in range(ec['batches']):
        text, spec, target, vid = proc.make_batch(ec['B'], P, case['n_words'], case['n_spk'], ec['seed'] + i)
        batches.append((None, None, torch.from_numpy(text), None, torch.from_numpy(target), torch.zeros(ec['B'], 1), torch.from_numpy(spec), None))
    es = proc.EpsStream(ec['seed'])
    for g_ in gens:
        g_.eps_source = lambda shape, device: torch.from_numpy(es(shape)).to(device)
    draws = iter(proc.eval_speakers(ec['B'] * ec['batches'], case['n_spk'], ec['seed']))
    perm, torch.randperm = torch.randperm, (lambda n, *a, **k: torch.arange(n - 1, -1, -1))
    try:
        ret = evaluate_testset(batches, gens, aud, ev, args, vid_source=lambda spk, n: [next(draws) for _ in range(n)])
    finally:
        torch.randperm = perm
    for k, rtol in (('loss', 1e-4), ('joint_mae', 1e-4), ('frechet', 1e-3), ('feat_dist', 1e-4), ('diversity', 1e-4)):
        ref = float(g[tag + '/' + k])
        tol = rtol * abs(ref) + 3 * max(float(g['%s/%s@noise' % (tag, k)]), float(g['%s/%s@cond' % (tag, k)]))
        assert abs(float(ret[k]) - ref) <= tol, (k, float(ret[k]), ref, tol)
    assert ret['bc'] == 0 and ret['_accel'] > 0
    assert all(m.training for m in gens) and aud.training                # back in training mode, train.py:475-479

"""FGD evaluator (scripts/model/embedding_space_evaluator.py:57-154) against tests/golden/fgd.npz -- the outputs of the
reference's own EmbeddingSpaceEvaluator (EmbeddingNet 'pose' mode auto-encoder with procedural weights) on three batches."""
import numpy as np
import pytest
import torch

from ha2g_amd import procedural as proc
from ha2g_amd.config import FGD_CASE, hierarchy_args
from ha2g_testing import fgd_ae_state as ae_state


def _close(got, g, key, rtol, scale=None, nm=3.0):
    ref = g[key]
    fl = max(float(g[key + '@noise']), float(g[key + '@cond']))
    sc = scale if scale is not None else max(np.abs(ref).max(), 1e-30)
    err = np.abs(np.asarray(got, np.float64) - ref).max()
    assert err <= rtol * sc + nm * fl, (key, err, rtol * sc + nm * fl)


def _trace_scale(g, tag='fgd'):
    feats = lambda p: np.vstack([g['%s/%s_feat%d' % (tag, p, i)] for i in range(FGD_CASE['batches'])])
    return float(np.trace(np.cov(feats('real'), rowvar=False)) + np.trace(np.cov(feats('gen'), rowvar=False)))


@pytest.mark.parametrize('dt', [torch.float64, torch.float32])
@pytest.mark.parametrize('P', [27, 126])
def test_oracle_fgd_matches_reference(golden, P, dt):
    from oracle import fgd_oracle as F
    g = golden('fgd')
    tag = 'fgd' if P == 27 else 'fgd126'
    ev = F.Evaluator(ae_state(dt, P)) if P == 27 else F.Evaluator(ae_state(dt, P), enc='encoder.', mu=None)
    rtol, nm = (1e-9, 0.0) if dt == torch.float64 else (1e-4, 3.0)
    for i in range(FGD_CASE['batches']):
        real, gen = proc.fgd_batch(FGD_CASE['B'], i, FGD_CASE['seed'], P=P)
        ev.push_samples(torch.from_numpy(gen).to(dt), torch.from_numpy(real).to(dt))
        _close(ev.real[-1], g, '%s/real_feat%d' % (tag, i), rtol, nm=nm)
        _close(ev.gen[-1], g, '%s/gen_feat%d' % (tag, i), rtol, nm=nm)
        _close(ev.recon_err_diff[-1], g, '%s/recon_err_diff%d' % (tag, i), rtol, nm=nm, scale=2.0)
        _close(ev.cos_err_diff[-1], g, '%s/cos_err_diff%d' % (tag, i), rtol, nm=nm)
    fd, dist = ev.get_scores()
    _close(fd, g, tag + '/frechet', rtol, scale=_trace_scale(g, tag), nm=nm)
    _close(dist, g, tag + '/feat_dist', rtol, nm=nm)


@pytest.mark.gpu
@pytest.mark.parametrize('P', [27, 126])
def test_gpu_fgd_evaluator_matches_reference(golden, P):
    from ha2g_amd.embedding_space_evaluator import EmbeddingSpaceEvaluator
    g = golden('fgd')
    tag = 'fgd' if P == 27 else 'fgd126'
    ckpt = {'pose_dim': 27, 'gen_dict': ae_state()} if P == 27 else {'pose_dim': 126, 'latent_dim': 128, 'motion_ae': ae_state(P=126)}
    ev = EmbeddingSpaceEvaluator(hierarchy_args(expressive=P == 126), ckpt, None, 'cuda:0')
    for i in range(FGD_CASE['batches']):
        real, gen = proc.fgd_batch(FGD_CASE['B'], i, FGD_CASE['seed'], P=P)
        ev.push_samples(None, None, torch.from_numpy(gen).cuda(), torch.from_numpy(real).cuda())
        _close(ev.real_feat_list[-1].cpu().numpy(), g, '%s/real_feat%d' % (tag, i), 1e-4)
        _close(ev.generated_feat_list[-1].cpu().numpy(), g, '%s/gen_feat%d' % (tag, i), 1e-4)
        _close(float(ev.recon_err_diff[-1]), g, '%s/recon_err_diff%d' % (tag, i), 1e-4, scale=2.0)
        _close(float(ev.cos_err_diff[-1]), g, '%s/cos_err_diff%d' % (tag, i), 1e-4)
    assert ev.get_no_of_samples() == FGD_CASE['batches']
    fd, dist = ev.get_scores()
    _close(fd, g, tag + '/frechet', 1e-4, scale=_trace_scale(g, tag))
    _close(dist, g, tag + '/feat_dist', 1e-4)
    assert ev.get_diversity_scores() >= 0          # a random re-pairing of 3 batches can be the identity
    ev.reset()
    assert ev.get_no_of_samples() == 0

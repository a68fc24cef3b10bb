"""FGD evaluator (scripts/model/embedding_space_evaluator.py:57-154) against tests/golden/fgd.npz -- the outputs of the
reference's own EmbeddingSpaceEvaluator (EmbeddingNet 'pose' mode auto-encoder with procedural weights) on three batches."""
import numpy as np
import pytest
import torch

from ha2g_amd import procedural as proc
from ha2g_amd.config import FGD_CASE, hierarchy_args


def ae_state(dt=torch.float32, P=27):
    """Procedural state dict of the reference's EmbeddingNet(mode='pose') (P=27; keys/shapes restated from model/embedding_net.py:42-82,
    156-218) or MotionAE(126, 128) (P=126; model/motion_ae.py:33-131)."""
    sch = {}
    latent = 32 if P == 27 else 128

    def bn(p, c):
        for k in ('weight', 'bias', 'running_mean', 'running_var'):
            sch[p + k] = (c,)
        sch[p + 'num_batches_tracked'] = ()
    e, d = ('pose_encoder.' if P == 27 else 'encoder.'), 'decoder.'
    for i, (co, ci, k) in enumerate(((32, P, 3), (64, 32, 3), (64, 64, 4))):
        sch['%snet.%d.0.weight' % (e, i)] = (co, ci, k); sch['%snet.%d.0.bias' % (e, i)] = (co,); bn('%snet.%d.1.' % (e, i), co)
    sch[e + 'net.3.weight'] = (32, 64, 3); sch[e + 'net.3.bias'] = (32,)
    for i, (co, ci) in ((0, (256, 384)), (3, (128, 256)), (6, (latent, 128))):
        sch['%sout_net.%d.weight' % (e, i)] = (co, ci); sch['%sout_net.%d.bias' % (e, i)] = (co,)
    bn(e + 'out_net.1.', 256); bn(e + 'out_net.4.', 128)
    if P == 27:
        for k in ('fc_mu', 'fc_logvar'):
            sch[e + k + '.weight'] = (32, 32); sch[e + k + '.bias'] = (32,)
    sch[d + 'pre_net.0.weight'] = (64, latent); sch[d + 'pre_net.0.bias'] = (64,); bn(d + 'pre_net.1.', 64)
    sch[d + 'pre_net.3.weight'] = (136, 64); sch[d + 'pre_net.3.bias'] = (136,)
    sch[d + 'net.0.weight'] = (4, 32, 3); sch[d + 'net.0.bias'] = (32,); bn(d + 'net.1.', 32)
    sch[d + 'net.3.weight'] = (32, 32, 3); sch[d + 'net.3.bias'] = (32,); bn(d + 'net.4.', 32)
    sch[d + 'net.6.weight'] = (32, 32, 3); sch[d + 'net.6.bias'] = (32,)
    sch[d + 'net.7.weight'] = (P, 32, 3); sch[d + 'net.7.bias'] = (P,)
    sd = {}
    for k, shp in sch.items():
        v = proc.tensor_for(('fgd.' if P == 27 else 'fgd126.') + k, shp if shp else (1,), FGD_CASE['seed'])
        t = torch.from_numpy(np.asarray(v)).reshape(shp)
        sd[k] = t.to(dt) if t.is_floating_point() else t
    return sd


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

"""Pin oracle/ha2g_oracle.py to fixtures produced by the reference itself (tests/golden/gen_golden.py).

CPU only.  The fixtures hold the reference's float64 results ("truth") and, per array, `@noise` = how
far the reference's own float32 run lands from that truth.  Two pins:
  * oracle in float64 == truth to 1e-9 rel  -> the restated ALGORITHM is the reference's;
  * oracle in float32 within 1e-4 rel + 3 x the reference's own float32 scatter (4 runs) -> the tolerance policy
    the GPU parity tests use (deep-net fp32 gradients are ill-conditioned: the reference's own fp32
    run is up to ~1e-2 rel away from its fp64 run on some ResNet gradients at B=3).
"""
import numpy as np
import pytest
import torch

from ha2g_amd import procedural as proc
from ha2g_amd import schema
from ha2g_amd.config import CASES, make_args
from ha2g_testing import Checker, batch_for, leaf_params, state_for, wproc
from oracle import ha2g_oracle as O

DTS = [torch.float64, torch.float32]


@pytest.mark.parametrize('dt', DTS)
@pytest.mark.parametrize('name', ['small', 'cfg1'])
def test_text_encoder(golden, name, dt):
    case, ck = CASES[name], Checker(golden(name), dt)
    sd = state_for(case, dt)
    text, _, _, _ = batch_for(case, dt)
    ps = leaf_params(sd, 'text')
    y = O.text_encoder_tcn(text, sd, 'text.', case['n_layers'])
    ck.close(y, 'text/out')
    grads = torch.autograd.grad((y * wproc('text', y, case['seed'])).sum(), list(ps.values()))
    ck.grads('text/grad', 'text', ps, grads)


@pytest.mark.parametrize('dt', DTS)
def test_wav_encoder(golden, dt):
    case, ck = CASES['small'], Checker(golden('small'), dt)
    sd = state_for(case, dt)
    _, spec, _, vid = batch_for(case, dt)
    ps = leaf_params(sd, 'audio')
    w, lo, mid, hi, blend = O.wav_encoder(spec, vid, sd, 'audio.', 3)
    ck.close(w, 'audio/weight')
    ck.close(lo, 'audio/low')
    ck.close(mid, 'audio/mid')
    ck.close(hi, 'audio/high')
    for i, b in enumerate(blend):
        ck.close(b, 'audio/blend%d' % i)
    s = case['seed']
    loss = sum((b * wproc('blend%d' % i, b, s)).sum() for i, b in enumerate(blend)) + (hi * wproc('hi', hi, s)).sum() \
        + (lo * wproc('lo', lo, s)).sum()
    grads = torch.autograd.grad(loss, list(ps.values()))
    ck.grads('audio/grad', 'audio', ps, grads)
    for k in sd:
        if k.startswith('audio.') and k.endswith(('running_mean', 'running_var')):
            ck.digest(sd[k], 'audio/buf/' + k[6:])


@pytest.mark.parametrize('dt', DTS)
@pytest.mark.parametrize('name', ['small', 'cfg1'])
def test_generator(golden, name, dt):
    case, ck = CASES[name], Checker(golden(name), dt)
    sd = state_for(case, dt)
    text, _, target, vid = batch_for(case, dt)
    B = case['B']
    ps = leaf_params(sd, 'g3')
    eps = torch.from_numpy(proc.EpsStream(case['seed'])((B, 16))).to(dt)
    pre = torch.zeros(B, 34, 28, dtype=dt)
    pre[:, :4, :-1] = target[:, :4]
    pre[:, :4, -1] = 1
    pre.requires_grad_(True)
    afeat = torch.from_numpy(proc.tensor_for('in.afeat', (B, 34, 32), case['seed']) * 10).to(dt).requires_grad_(True)
    o, z, mu, lv = O.pose_generator(pre, text, afeat, vid, sd, 'g3.', case['n_layers'], case['hidden_size'], eps)
    ck.close(o, 'gen/out')
    ck.close(z, 'gen/z')
    ck.close(mu, 'gen/mu')
    ck.close(lv, 'gen/logvar')
    s = case['seed']
    loss = (o * wproc('gen', o, s)).sum() + (z * wproc('z', z, s)).sum() + (mu * lv).sum()
    grads = torch.autograd.grad(loss, list(ps.values()) + [pre, afeat])
    ck.grads('gen/grad', 'g3', ps, grads[:-2])
    ck.close(grads[-2], 'gen/grad_pre')
    ck.close(grads[-1], 'gen/grad_afeat')


@pytest.mark.parametrize('dt', DTS)
@pytest.mark.parametrize('name', ['small', 'cfg1'])
def test_discriminator(golden, name, dt):
    case, ck = CASES[name], Checker(golden(name), dt)
    sd = state_for(case, dt)
    _, _, target, _ = batch_for(case, dt)
    ps = leaf_params(sd, 'dis')
    x = target.clone().requires_grad_(True)
    d = O.conv_discriminator(x, sd, 'dis.')
    ck.close(d, 'dis/out')
    grads = torch.autograd.grad((d * wproc('dis', d, case['seed'])).sum(), list(ps.values()) + [x])
    ck.grads('dis/grad', 'dis', ps, grads[:-1])
    ck.close(grads[-1], 'dis/grad_in')
    for k in sd:
        if k.startswith('dis.') and k.endswith(('running_mean', 'running_var')):
            ck.close(sd[k], 'dis/buf/' + k[4:])


@pytest.mark.parametrize('dt', DTS)
@pytest.mark.parametrize('name', ['small', 'cfg1'])
@pytest.mark.parametrize('expr', [False, True])
def test_contrastive(golden, name, expr, dt):
    case, ck = CASES[name], Checker(golden(name), dt)
    N = case['B'] * 34
    a = torch.from_numpy(proc.tensor_for('in.ca', (N, 32), case['seed']) * 6).to(dt).requires_grad_(True)
    b = torch.from_numpy(proc.tensor_for('in.cb', (N, 32), case['seed']) * 6).to(dt).requires_grad_(True)
    l = O.contrastive_ce(a, b, expressive=expr, chunk=37)
    ga, gb = torch.autograd.grad(l, [a, b])
    tag = 'contrastive_expr' if expr else 'contrastive'
    ck.close(l, tag + '/loss')
    ck.close(ga, tag + '/grad_a')
    ck.close(gb, tag + '/grad_b')


@pytest.mark.parametrize('dt', DTS)
@pytest.mark.parametrize('name', ['small', 'cfg1'])
def test_train_step(golden, name, dt):
    """Two consecutive steps (epoch 0, 11): loss dicts, accumulated grads, Adam-updated params, BN stats."""
    case, g = CASES[name], golden(name)
    ck = Checker(g, dt)
    sd = state_for(case, dt)
    text, spec, target, vid = batch_for(case, dt)
    args = make_args(case)
    tr = O.OracleTrainer(sd, args)
    es = proc.EpsStream(case['seed'])
    perm = torch.from_numpy(proc.fixed_perm(case['B'], case['seed']))
    for si, epoch in enumerate((0, 11)):
        ret = tr.train_iter(epoch, text, spec, target, vid, lambda shp: torch.from_numpy(es(shp)).to(dt), perm)
        ck.step(si, ret, tr.grads, sd)


@pytest.mark.parametrize('dt', DTS)
@pytest.mark.parametrize('name', ['expr_small', 'expr_cfg1'])
def test_train_step_expressive(golden, name, dt):
    """6-level TED-Expressive twin (train_hierarchy_expressive.py): off-by-one head scatter, palm normals, eps-free contrastive."""
    from ha2g_amd.config import EXPRESSIVE_SPEC
    case, g = CASES[name], golden(name)
    ck = Checker(g, dt)
    sd = state_for(case, dt, schema.EXPRESSIVE_POSE_DIMS)
    text, spec, target, vid = batch_for(case, dt, P=126)
    args = make_args(case)
    tr = O.OracleTrainer(sd, args, EXPRESSIVE_SPEC)
    es = proc.EpsStream(case['seed'])
    perm = torch.from_numpy(proc.fixed_perm(case['B'], case['seed']))
    for si, epoch in enumerate((0, 11)):
        ret = tr.train_iter(epoch, text, spec, target, vid, lambda shp: torch.from_numpy(es(shp)).to(dt), perm)
        ck.step(si, ret, tr.grads, sd)


@pytest.mark.parametrize('dt', DTS)
@pytest.mark.parametrize('name', ['small', 'cfg1'])
def test_eval_mode_inference(golden, name, dt):
    """Inference path (module.eval(): BatchNorm running statistics) of audio encoder, discriminator and generator."""
    case, ck = CASES[name], Checker(golden(name), dt)
    sd = state_for(case, dt)
    text, spec, target, vid = batch_for(case, dt)
    B = case['B']
    with torch.no_grad():
        w, lo, mid, hi, blend = O.wav_encoder(spec, vid, sd, 'audio.', 3, update_bn='eval')
        ck.close(lo, 'eval/audio/low')
        ck.close(hi, 'eval/audio/high')
        ck.close(blend[2], 'eval/audio/blend2')
        ck.close(O.conv_discriminator(target, sd, 'dis.', update_bn='eval'), 'eval/dis/out')
        eps = torch.from_numpy(proc.EpsStream(case['seed'])((B, 16))).to(dt)
        pre = torch.zeros(B, 34, 28, dtype=dt)
        pre[:, :4, :-1] = target[:, :4]
        pre[:, :4, -1] = 1
        o, *_ = O.pose_generator(pre, text, blend[2], vid, sd, 'g3.', case['n_layers'], case['hidden_size'], eps)
        ck.close(o, 'eval/gen/out')


@pytest.mark.parametrize('dt', DTS)
@pytest.mark.parametrize('name', ['cfg1_gan', 'expr_cfg1_gan'])
def test_gan_phase_first_step(golden, name, dt):
    """The `*_gan` fixtures: ONE GAN-phase step (epoch 11) from fresh state run by the reference -- the oracle's D phase, gen_error path, D-gradient
    accumulation through the updated D and the six (nine) Adam updates under the strict step-0 policy (1e-9 in float64)."""
    from ha2g_amd.config import EXPRESSIVE_SPEC
    case, g = CASES[name[:-4]], golden(name)
    expressive = bool(case.get('expressive'))
    ck = Checker(g, dt)
    sd = state_for(case, dt, schema.EXPRESSIVE_POSE_DIMS if expressive else schema.GESTURE_POSE_DIMS)
    text, spec, target, vid = batch_for(case, dt, P=126 if expressive else 27)
    tr = O.OracleTrainer(sd, make_args(case), EXPRESSIVE_SPEC if expressive else None)
    es = proc.EpsStream(case['seed'])
    perm = torch.from_numpy(proc.fixed_perm(case['B'], case['seed']))
    ret = tr.train_iter(11, text, spec, target, vid, lambda shp: torch.from_numpy(es(shp)).to(dt), perm)
    assert 'gen' in ret and 'dis' in ret and any(k.startswith('dis.') for k in tr.grads)
    ck.step(0, ret, tr.grads, sd)


def test_imposing_the_oracles_own_activation_sequence_reproduces_the_step():
    """oracle.act_sequence + relu_pattern (the whole-step linearisation of tests/test_gpu_linearised.py): with the decisions the float64 oracle itself took
    imposed -- the tower's by name, the text encoders' / heads' / discriminator's in call order -- a GAN-phase step gives the same losses and gradients."""
    case = CASES['small']
    dt = torch.float64
    text, spec, target, vid = batch_for(case, dt)
    perm = torch.from_numpy(proc.fixed_perm(case['B'], case['seed']))

    def run(masks=None, seq=None, rec_m=None, rec_s=None):
        sd = state_for(case, dt)
        tr = O.OracleTrainer(sd, make_args(case))
        es = proc.EpsStream(case['seed'])
        with O.relu_pattern(masks=masks, record=rec_m), O.act_sequence(masks=seq, record=rec_s):
            ret = tr.train_iter(11, text, spec, target, vid, lambda shp: torch.from_numpy(es(shp)).to(dt), perm)
        return ret, tr.grads
    rec_m, rec_s = {}, []
    r0, g0 = run(rec_m=rec_m, rec_s=rec_s)
    # per generator call: 2 layers x (2 conv ReLUs + 1 residual ReLU) in its text encoder + 1 head LeakyReLU; 9 generator calls, 1 stand-alone encoder, 3 D calls x 2
    assert len(rec_s) == 6 + 9 * 7 + 3 * 2 and len(rec_m) == 1 + 16 * 3 + 3
    r1, g1 = run(masks=rec_m, seq=rec_s)
    assert r0.keys() == r1.keys() and all(abs(r0[k] - r1[k]) <= 1e-12 * max(1.0, abs(r0[k])) for k in r0)
    for k in g0:
        assert torch.allclose(g0[k], g1[k], rtol=1e-12, atol=1e-14), k

"""GPU parity of the module mirrors (HIP forward + hand-written backward) against the reference-generated
fixtures, with the tolerance policy of tests/ha2g_testing.py (1e-4 rel + 3x the reference's own fp32 scatter)."""
import numpy as np
import pytest
import torch

from ha2g_amd import procedural as proc
from ha2g_amd.config import CASES, EXPRESSIVE_SPEC, MEAN_DIR_VEC_EXPRESSIVE
from ha2g_testing import Checker, batch_for, build_modules, wproc

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def grads_of(module, ck, prefix):
    for k, p in module.named_parameters():
        if '.net.' in k:
            continue
        assert p.grad is not None, k
        ck.digest(p.grad, '%s/%s' % (prefix, k))


@pytest.mark.parametrize('name', ['small', 'cfg1'])
def test_text_encoder(golden, name):
    case, ck = CASES[name], Checker(golden(name))
    _, _, _, _, txt = build_modules(case, DEV)
    text, _, _, _ = batch_for(case)
    y = txt(text.to(DEV))
    ck.close(y, 'text/out')
    (y * wproc('text', y, case['seed'])).sum().backward()
    grads_of(txt, ck, 'text/grad')


@pytest.mark.parametrize('name', ['small', 'cfg1'])
def test_generator(golden, name):
    case, ck = CASES[name], Checker(golden(name))
    _, gens, _, _, _ = build_modules(case, DEV)
    g3 = gens[2]
    text, _, target, vid = batch_for(case)
    B = case['B']
    eps = torch.from_numpy(proc.EpsStream(case['seed'])((B, 16))).to(DEV)
    g3.eps_source = lambda shape, device: eps
    pre = torch.zeros(B, 34, 28)
    pre[:, :4, :-1] = target[:, :4]
    pre[:, :4, -1] = 1
    pre = pre.to(DEV).requires_grad_(True)
    afeat = torch.from_numpy(proc.tensor_for('in.afeat', (B, 34, 32), case['seed']) * 10).to(DEV).requires_grad_(True)
    o, z, mu, lv = g3(pre, text.to(DEV), afeat, vid.to(DEV))
    ck.close(o, 'gen/out')
    ck.close(z, 'gen/z')
    ck.close(mu, 'gen/mu')
    ck.close(lv, 'gen/logvar')
    s = case['seed']
    ((o * wproc('gen', o, s)).sum() + (z * wproc('z', z, s)).sum() + (mu * lv).sum()).backward()
    grads_of(g3, ck, 'gen/grad')
    ck.close(pre.grad, 'gen/grad_pre')
    ck.close(afeat.grad, 'gen/grad_afeat')


@pytest.mark.parametrize('name', ['small', 'cfg1'])
def test_discriminator(golden, name):
    case, ck = CASES[name], Checker(golden(name))
    _, _, dis, _, _ = build_modules(case, DEV)
    _, _, target, _ = batch_for(case)
    x = target.to(DEV).requires_grad_(True)
    d = dis(x)
    ck.close(d, 'dis/out')
    (d * wproc('dis', d, case['seed'])).sum().backward()
    grads_of(dis, ck, 'dis/grad')
    ck.close(x.grad, 'dis/grad_in')
    for k, b in dis.named_buffers():
        if k.endswith(('running_mean', 'running_var')):
            ck.close(b, 'dis/buf/' + k)


def test_wav_encoder(golden):
    case, ck = CASES['small'], Checker(golden('small'))
    _, _, _, aud, _ = build_modules(case, DEV)
    _, spec, _, vid = batch_for(case)
    w, lo, mid, hi, blend = aud(spec.to(DEV), vid.to(DEV))
    ck.close(w, 'audio/weight')
    ck.close(lo, 'audio/low')
    ck.close(mid, 'audio/mid')
    ck.close(hi, 'audio/high')
    for i, b in enumerate(blend):
        ck.close(b, 'audio/blend%d' % i)
    s = case['seed']
    loss = sum((b * wproc('blend%d' % i, b, s)).sum() for i, b in enumerate(blend)) + (hi * wproc('hi', hi, s)).sum() \
        + (lo * wproc('lo', lo, s)).sum()
    loss.backward()
    grads_of(aud, ck, 'audio/grad')
    for k, b in aud.named_buffers():
        if k.endswith(('running_mean', 'running_var')):
            ck.digest(b, 'audio/buf/' + k)


@pytest.mark.parametrize('name', ['small', 'cfg1'])
@pytest.mark.parametrize('expr', [False, True])
def test_contrastive(golden, name, expr):
    from ha2g_amd import ops
    case, ck = CASES[name], Checker(golden(name))
    N = case['B'] * 34
    a = torch.from_numpy(proc.tensor_for('in.ca', (N, 32), case['seed']) * 6).to(DEV).requires_grad_(True)
    b = torch.from_numpy(proc.tensor_for('in.cb', (N, 32), case['seed']) * 6).to(DEV).requires_grad_(True)
    l = ops.contrastive(a, b, expr)
    l.backward()
    tag = 'contrastive_expr' if expr else 'contrastive'
    ck.close(l, tag + '/loss')
    ck.close(a.grad, tag + '/grad_a')
    ck.close(b.grad, tag + '/grad_b')


def test_losses_vs_torch():
    """Small loss kernels against the oracle's float64 formulas (value and gradient)."""
    from ha2g_amd import ops
    from ha2g_amd.config import PHYS_GESTURE, PHYS_GESTURE_PAIRS, MEAN_DIR_VEC_GESTURE, EXPRESSIVE_SPEC, MEAN_DIR_VEC_EXPRESSIVE
    from oracle import ha2g_oracle as O
    r = np.random.Generator(np.random.PCG64(5))
    B, T, P = 6, 34, 27
    out = torch.from_numpy((0.3 * r.standard_normal((B, T, P))).astype(np.float32))
    tgt = torch.from_numpy((0.3 * r.standard_normal((B, T, P))).astype(np.float32))
    z, zr = torch.from_numpy(r.standard_normal((B, 16)).astype(np.float32)), torch.from_numpy(r.standard_normal((B, 16)).astype(np.float32))
    zr[0] = z[0] + 1e-7                                                  # forces the clamp(min=-1000) branch
    mu, lv = torch.from_numpy(r.standard_normal((B, 16)).astype(np.float32)), torch.from_numpy(r.standard_normal((B, 16)).astype(np.float32))
    dprob = torch.from_numpy(r.random((B, 1)).astype(np.float32))
    dprob2 = torch.from_numpy(r.random((B, 1)).astype(np.float32))
    mdv = torch.tensor(MEAN_DIR_VEC_GESTURE)

    def ref():
        o = out.double().requires_grad_(True)
        m, l = mu.double().requires_grad_(True), lv.double().requires_grad_(True)
        d1, d2 = dprob.double().requires_grad_(True), dprob2.double().requires_grad_(True)
        hub = O.huber(o, tgt.double(), 0.1)
        pose = O.huber(o, tgt.double(), 0.05, 'none').sum((1, 2))
        zl1 = (z.double() - zr.double()).abs().mean(1)
        div = torch.clamp(-(pose / (zl1 + 1e-5)), min=-1000).mean()
        kl = -0.5 * torch.mean(1 + l - m.pow(2) - l.exp())
        phy = O.physical_prior(o, mdv.double(), PHYS_GESTURE_PAIRS, PHYS_GESTURE[0], PHYS_GESTURE[1])
        gen = -torch.mean(torch.log(d1 + 1e-8))
        dis = -torch.mean(torch.log(d1 + 1e-8) + torch.log(1 - d2 + 1e-8))
        vals = [hub, div, kl, phy, gen, dis]
        tot = 1.5 * hub + 0.7 * div + 0.3 * kl + 2.0 * phy + 0.9 * gen + 1.1 * dis
        g = torch.autograd.grad(tot, [o, m, l, d1, d2])
        return [float(v) for v in vals], g

    vals64, g64 = ref()
    o = out.to(DEV).requires_grad_(True)
    m, l = mu.to(DEV).requires_grad_(True), lv.to(DEV).requires_grad_(True)
    d1, d2 = dprob.to(DEV).requires_grad_(True), dprob2.to(DEV).requires_grad_(True)
    pairs = torch.tensor(PHYS_GESTURE_PAIRS, dtype=torch.int32, device=DEV)
    avg, var = torch.tensor(PHYS_GESTURE[0], device=DEV), torch.tensor(PHYS_GESTURE[1], device=DEV)
    vals = [ops.huber(o, tgt.to(DEV), 0.1), ops.div_reg(o, tgt.to(DEV), z.to(DEV), zr.to(DEV), 0.05), ops.kld(m, l),
            ops.phys_angle(o, mdv.to(DEV), pairs, avg, var), ops.gen_loss(d1), ops.dis_loss(d1, d2)]
    tot = 1.5 * vals[0] + 0.7 * vals[1] + 0.3 * vals[2] + 2.0 * vals[3] + 0.9 * vals[4] + 1.1 * vals[5]
    tot.backward()
    for v, r64 in zip(vals, vals64):
        assert abs(float(v) - r64) <= 2e-5 * max(abs(r64), 1e-3), (float(v), r64)
    for got, r64 in zip([o.grad, m.grad, l.grad, d1.grad, d2.grad], g64):
        err = float((got.double().cpu() - r64).abs().max() / r64.abs().max())
        assert err < 5e-5, err


def test_phys_angle_expressive_palm():
    """41 angle pairs over 42 bones + two palm normals (cross products) vs the oracle's float64 formula."""
    from ha2g_amd import ops
    from oracle import ha2g_oracle as O
    sp = EXPRESSIVE_SPEC
    r = np.random.Generator(np.random.PCG64(9))
    out = torch.from_numpy((0.3 * r.standard_normal((5, 34, 126))).astype(np.float32))
    mdv = torch.tensor(MEAN_DIR_VEC_EXPRESSIVE)
    o64 = out.double().requires_grad_(True)
    l64 = O.physical_prior(o64, mdv.double(), sp['phys_pairs'], sp['phys_avg'], sp['phys_var'], sp['palm'])
    g64, = torch.autograd.grad(l64, o64)
    o = out.to(DEV).requires_grad_(True)
    l = ops.phys_angle(o, mdv.to(DEV), torch.tensor(sp['phys_pairs'], dtype=torch.int32, device=DEV),
                       torch.tensor(sp['phys_avg'], device=DEV), torch.tensor(sp['phys_var'], device=DEV), sp['palm'])
    (2.0 * l).backward()
    assert abs(float(l) - float(l64)) <= 2e-5 * abs(float(l64))
    err = float((o.grad.double().cpu() - 2.0 * g64).abs().max() / (2.0 * g64).abs().max())
    assert err < 5e-5, err


@pytest.mark.parametrize('name', ['small', 'cfg1'])
def test_eval_mode_inference(golden, name):
    """module.eval() forward (BatchNorm running statistics, no dropout) = the path evaluate_testset / synthesis use
    (reference scripts/train.py:376-417)."""
    case, ck = CASES[name], Checker(golden(name))
    _, gens, dis, aud, _ = build_modules(case, DEV)
    for m in gens + [dis, aud]:
        m.eval()
    text, spec, target, vid = (t.to(DEV) for t in batch_for(case))
    B = case['B']
    eps = torch.from_numpy(proc.EpsStream(case['seed'])((B, 16))).to(DEV)
    gens[2].eps_source = lambda shape, device: eps
    with torch.no_grad():
        w, lo, mid, hi, blend = aud(spec, vid)
        ck.close(lo, 'eval/audio/low')
        ck.close(hi, 'eval/audio/high')
        ck.close(blend[2], 'eval/audio/blend2')
        ck.close(dis(target), 'eval/dis/out')
        pre = torch.zeros(B, 34, 28, device=DEV)
        pre[:, :4, :-1] = target[:, :4]
        pre[:, :4, -1] = 1
        o, *_ = gens[2](pre, text, blend[2], vid)
        ck.close(o, 'eval/gen/out')
    for k, b in aud.named_buffers():                      # eval must not touch the running statistics
        if k.endswith('running_mean'):
            ref = torch.from_numpy(proc.tensor_for('audio.' + k, b.shape, case['seed']))
            assert torch.equal(b.cpu(), ref)

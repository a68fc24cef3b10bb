"""GPU parity of individual HIP kernels (through the C-ABI) against float64 CPU restatements.
Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), 'GPU tests need a MI355X'
    return torch.device('cuda:0')


def rnd(shape, seed, scale=1.0):
    r = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((scale * r.standard_normal(shape)).astype(np.float32))


def relerr(got, ref):
    ref = ref.double()
    return float((got.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


GEMM_CASES = [
    # (M, N, K, transa, transb, bias, act, beta)
    (4352, 900, 108, False, True, True, 0, 0.0),      # GRU input projection, layer 0
    (4352, 900, 102, False, True, True, 0, 0.0),      # K not a multiple of 4 -> scalar path
    (4352, 900, 600, False, True, True, 0, 0.0),
    (136, 96, 64, False, True, True, 0, 0.0),
    (4352, 600, 900, False, False, False, 0, 1.0),    # dX = dgi W_ih, accumulate
    (900, 600, 4352, True, False, False, 0, 0.0),     # dW = dgi^T X  (split-K)
    (900, 102, 4352, True, False, False, 0, 1.0),     # unaligned N
    (4352, 150, 300, False, True, True, 2, 0.0),      # head Linear + LeakyReLU
    (4352, 27, 150, False, True, True, 0, 0.0),       # skinny N
    (4352, 32, 4032, False, True, True, 0, 0.0),      # tap FC
    (32, 4032, 4352, True, False, False, 0, 0.0),
    (7, 5, 3, False, True, False, 1, 0.0),
    (128, 16, 16, False, True, True, 3, 0.0),         # sigmoid epilogue
    (300, 300, 17, False, False, True, 1, 0.5),
]


@pytest.mark.parametrize('case', GEMM_CASES)
def test_gemm(case):
    from ha2g_amd import ops
    M, N, K, ta, tb, use_bias, act, beta = case
    dev = _dev()
    a = rnd((K, M) if ta else (M, K), 1)
    b = rnd((N, K) if tb else (K, N), 2)
    c0 = rnd((M, N), 3)
    bias = rnd((N,), 4) if use_bias else None
    ref = (a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double())
    if bias is not None:
        ref = ref + bias.double()
    ref = ref + beta * c0.double()
    if act == 1:
        ref = ref.clamp_min(0)
    elif act == 2:
        ref = torch.where(ref > 0, ref, 0.01 * ref)
    elif act == 3:
        ref = torch.sigmoid(ref)
    out = c0.to(dev).clone()
    ops.gemm(a.to(dev), b.to(dev), ta, tb, out=out, beta=beta, bias=None if bias is None else bias.to(dev), act=act)
    assert relerr(out, ref) < 2e-6 * max(1.0, K ** 0.5)


def test_gemm_strided_views():
    """Column slices of wider buffers as A, B and C (row stride != width)."""
    from ha2g_amd import ops
    dev = _dev()
    abuf, bbuf, cbuf = rnd((200, 64), 5).to(dev), rnd((96, 80), 6).to(dev), torch.zeros(200, 128, device=dev)
    a, b, c = abuf[:, 8:56], bbuf[:, 16:64], cbuf[:, 32:128]
    ops.gemm(a, b, transb=True, out=c)
    ref = a.double().cpu() @ b.double().cpu().t()
    assert relerr(c, ref) < 1e-5
    assert float(cbuf[:, :32].abs().max()) == 0.0


GRU_CASES = [
    # (B, T, In, H, L)
    (3, 34, 10, 32, 2),
    (20, 7, 8, 64, 4),
    (5, 34, 108, 300, 1),
    (37, 34, 102, 300, 2),
    (16, 28, 8, 64, 4),
]


@pytest.mark.parametrize('case', GRU_CASES)
def test_bigru_fwd_bwd(case):
    from ha2g_amd import ops
    from oracle import ha2g_oracle as O
    B, T, In, H, L = case
    dev = _dev()
    sd = {}
    flat = []
    for l in range(L):
        k = In if l == 0 else 2 * H
        for suf in ('', '_reverse'):
            for nm, shp in (('weight_ih', (3 * H, k)), ('weight_hh', (3 * H, H)), ('bias_ih', (3 * H,)), ('bias_hh', (3 * H,))):
                key = 'g.%s_l%d%s' % (nm, l, suf)
                sd[key] = rnd(shp, hash(key) % 100000, 1.0 / H ** 0.5)
                flat.append(key)
    x = rnd((B, T, In), 7)
    wy = rnd((B, T, 2 * H), 8)
    masks = [((rnd((B, T, 2 * H), 20 + l) > -0.5).float() / 0.7) for l in range(L - 1)]
    # float64 oracle
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    x64 = x.double().requires_grad_(True)
    y64 = O.gru_bidir(x64, sd64, 'g.', L, H, [m.double() for m in masks])
    g64 = torch.autograd.grad((y64 * wy.double()).sum(), [x64] + [sd64[k] for k in flat])
    # HIP
    xg = x.to(dev).requires_grad_(True)
    wg = [sd[k].to(dev).requires_grad_(True) for k in flat]
    y = ops.bigru(xg, wg, H, [m.to(dev) for m in masks])
    gg = torch.autograd.grad((y * wy.to(dev)).sum(), [xg] + wg)
    assert relerr(y, y64.detach()) < 2e-5
    for name, a, b in zip(['x'] + flat, gg, g64):
        assert relerr(a, b) < 1e-4, name

"""GPU parity of individual HIP kernels (through the C-ABI) against float64 CPU restatements.
Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""
import zlib

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
                sd[key] = rnd(shp, zlib.crc32(key.encode()) % 100000, 1.0 / H ** 0.5)      # stable across processes (str hash is salted)
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


CONV_CASES = [
    # (N, H, W, Cin, Cout, k, stride, pad, bias)
    (2, 16, 9, 256, 256, 3, 1, 1, False),
    (3, 32, 18, 128, 256, 3, 2, 1, False),
    (2, 32, 18, 128, 256, 1, 2, 0, False),
    (2, 64, 35, 64, 64, 2, 1, 0, True),
    (2, 64, 36, 16, 16, 3, 1, 0, True),
    (2, 40, 22, 32, 32, 3, 1, 1, False),
    (2, 41, 23, 32, 64, 3, 2, 1, False),
    (1, 128, 70, 32, 64, 1, 2, 0, False),
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv2d_fwd_dgrad_wgrad(case):
    """Implicit-GEMM NHWC conv kernels vs torch CPU float64 conv2d autograd."""
    import torch.nn.functional as F
    from ha2g_amd import wav_engine as we
    N, H, W, Cin, Cout, k, stride, pad, use_bias = case
    dev = _dev()
    x = rnd((N, Cin, H, W), 11)
    w = rnd((Cout, Cin, k, k), 12, 1.0 / (Cin * k * k) ** 0.5)
    b = rnd((Cout,), 13) if use_bias else None
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, None if b is None else b.double(), stride=stride, padding=pad)
    gy = rnd(tuple(y64.shape), 14)
    gx64, gw64 = torch.autograd.grad((y64 * gy.double()).sum(), [x64, w64])
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev)                      # NHWC
    wg = w.permute(0, 2, 3, 1).contiguous().to(dev)                      # OHWI
    y = we.conv_fwd(xg, wg, None if b is None else b.to(dev), stride, pad, 0)
    assert relerr(y.permute(0, 3, 1, 2), y64.detach()) < 1e-5
    gyg = gy.permute(0, 2, 3, 1).contiguous().to(dev)
    dx = we.conv_dgrad(gyg, wg, xg.shape, stride, pad)
    assert relerr(dx.permute(0, 3, 1, 2), gx64) < 1e-5
    dw = we.conv_wgrad(xg, gyg, wg, stride, pad)
    assert dw.shape == w.shape
    assert relerr(dw, gw64) < 2e-5


def test_stem_conv():
    import torch.nn.functional as F
    from ha2g_amd._lib import lib, check
    from ha2g_amd import ops
    dev = _dev()
    N, H, W = 3, 128, 70
    x = rnd((N, H, W), 21, 30.0)
    w = rnd((32, 1, 3, 3), 22, 0.3)
    b = rnd((32,), 23)
    x64, w64, b64 = x.double(), w.double().requires_grad_(True), b.double().requires_grad_(True)
    pre = F.conv2d(x64.unsqueeze(1), w64, b64, padding=1)
    gy = rnd(tuple(pre.shape), 24)
    gw64, gb64 = torch.autograd.grad((pre * gy.double()).sum(), [w64, b64])
    y = torch.empty(N, H, W, 32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    xg, wg, bg = x.to(dev), w.to(dev), b.to(dev)
    check(lib.ha2g_stem_conv_fwd_f32(xg.data_ptr(), wg.data_ptr(), bg.data_ptr(), y.data_ptr(), N, H, W, st))
    assert relerr(y.permute(0, 3, 1, 2), pre.detach().clamp_min(0)) < 1e-5
    dw, db = torch.empty(32, 1, 3, 3, device=dev), torch.empty(32, device=dev)
    gyg = gy.permute(0, 2, 3, 1).contiguous().to(dev)
    check(lib.ha2g_stem_conv_wgrad_f32(xg.data_ptr(), gyg.data_ptr(), dw.data_ptr(), db.data_ptr(), N, H, W, 0.0,
                                       ops.workspace(xg.device).data_ptr(), st))
    assert relerr(dw, gw64) < 2e-5
    assert relerr(db, gb64) < 2e-5


@pytest.mark.parametrize('shape', [(5000, 32), (96, 16), (1200, 256), (84, 8)])
def test_batchnorm_fwd_bwd(shape):
    import torch.nn.functional as F
    from ha2g_amd import ops
    dev = _dev()
    rows, C = shape
    x = rnd((rows, C), 31, 3.0) + 5.0                                    # mean >> 0: exercises the shifted variance
    g, b = rnd((C,), 32) + 1.5, rnd((C,), 33)
    rm, rv = rnd((C,), 34), rnd((C,), 35).abs() + 0.5
    wy = rnd((rows, C), 36)
    x64, g64, b64 = x.double().requires_grad_(True), g.double().requires_grad_(True), b.double().requires_grad_(True)
    rm64, rv64 = rm.double().clone(), rv.double().clone()
    y64 = F.batch_norm(x64, rm64, rv64, g64, b64, True, 0.1, 1e-5)
    gr = torch.autograd.grad((y64 * wy.double()).sum(), [x64, g64, b64])
    xg, gg, bg = x.to(dev).requires_grad_(True), g.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    rmg, rvg = rm.to(dev), rv.to(dev)
    y = ops.batch_norm_train(xg, gg, bg, rmg, rvg)
    (y * wy.to(dev)).sum().backward()
    assert relerr(y, y64.detach()) < 1e-5
    assert relerr(rmg, rm64) < 1e-5 and relerr(rvg, rv64) < 1e-5
    assert relerr(xg.grad, gr[0]) < 2e-5
    assert relerr(gg.grad, gr[1]) < 2e-5 and relerr(bg.grad, gr[2]) < 2e-5


@pytest.mark.parametrize('B', [5, 40, 384, 400])
def test_gru_cluster_matches_single_workgroup_kernel(B):
    """The 5-workgroup-cluster forward (weights in registers, per-step granule all-gather) must reproduce the
    single-workgroup kernel bit for bit (same MFMA order per output), for ragged, full and multi-launch batches."""
    from ha2g_amd import ops
    dev = _dev()
    H, T, In, L = 300, 34, 108, 2
    flat = []
    for l in range(L):
        k = In if l == 0 else 2 * H
        for suf in range(2):
            for shp in ((3 * H, k), (3 * H, H), (3 * H,), (3 * H,)):
                flat.append(rnd(shp, 100 + len(flat), 1.0 / H ** 0.5).to(dev))
    x = rnd((B, T, In), 7).to(dev)
    outs = []
    for use, fwd3 in ((False, False), (True, False), (True, True)):
        ops.USE_GRU_CLUSTER, old3 = use, ops.GRU_FWD3
        ops.GRU_FWD3 = fwd3
        try:
            outs.append(ops.bigru(x, flat, H))
        finally:
            ops.USE_GRU_CLUSTER, ops.GRU_FWD3 = True, old3
    assert ops.gru_cluster_error(dev) == 0
    assert torch.equal(outs[0], outs[1])                   # the fp32 cluster chain == the single-workgroup fp32 chain, bit for bit
    # the three-piece chain (round 4, the default in the fp32-class mode): the same function at fp32-class accuracy -- against float64 it is as
    # close as the fp32 chain is (two stacked layers, 34 steps of recurrence each)
    from oracle import ha2g_oracle as O
    names = []
    sd = {}
    it = iter(flat)
    for l in range(L):
        for suf in ('', '_reverse'):
            for nm in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh'):
                sd['g.%s_l%d%s' % (nm, l, suf)] = next(it).double().cpu()
    y64 = O.gru_bidir(x.double().cpu(), sd, 'g.', L, H)
    e32, e3 = relerr(outs[1], y64), relerr(outs[2], y64)
    assert ops.gru_fwd3_active(H, T) and not torch.equal(outs[1], outs[2])
    assert e3 < 2e-6 and e3 < 2.0 * e32 + 2e-7, (e3, e32)


@pytest.mark.parametrize('B', [5, 40, 130])
def test_gru_cluster_backward_matches(B):
    """Cluster BPTT vs the single-workgroup BPTT: same math, different (fixed) summation order of the five partial sums."""
    from ha2g_amd import ops
    dev = _dev()
    H, T, In, L = 300, 34, 108, 2
    ws = []
    for l in range(L):
        k = In if l == 0 else 2 * H
        for suf in range(2):
            for shp in ((3 * H, k), (3 * H, H), (3 * H,), (3 * H,)):
                ws.append(rnd(shp, 300 + len(ws), 1.0 / H ** 0.5).to(dev))
    x0 = rnd((B, T, In), 9).to(dev)
    wy = rnd((B, T, 2 * H), 10).to(dev)
    res = []
    for use, fwd3 in ((False, False), (True, False), (True, True)):       # gru.hip / fp32 cluster chains / three-piece cluster chains (round 4)
        ops.USE_GRU_CLUSTER, old3 = use, ops.GRU_FWD3
        ops.GRU_FWD3 = fwd3
        try:
            x = x0.clone().requires_grad_(True)
            flat = [w.clone().requires_grad_(True) for w in ws]
            yy = ops.bigru(x, flat, H)
            res.append(torch.autograd.grad((yy * wy).sum(), [x] + flat))
        finally:
            ops.USE_GRU_CLUSTER, ops.GRU_FWD3 = True, old3
    assert ops.gru_cluster_error(dev) == 0
    for a, b, c in zip(*res):
        assert relerr(a, b.cpu()) < 2e-5
        assert relerr(c, b.cpu()) < 2e-5                     # fp32-class: the same closeness to the fp32 chains as they have to each other


def test_split_bf16_core_opt_in():
    """The opt-in bf16x3 matrix core (3 x bf16 MFMA, fp32 accumulate) stays within 1e-5 rel of float64 on a GEMM and a conv."""
    import torch.nn.functional as F
    from ha2g_amd import ops, wav_engine as we
    from ha2g_amd._lib import lib
    dev = _dev()
    a, b = rnd((1000, 600), 41), rnd((300, 600), 42)
    x = rnd((2, 64, 20, 12), 43)
    w = rnd((64, 64, 3, 3), 44, 0.05)
    lib.ha2g_gemm_set_mode(7)
    try:
        c = ops.gemm(a.to(dev), b.to(dev), transb=True)
        y = we.conv_fwd(x.permute(0, 2, 3, 1).contiguous().to(dev), w.permute(0, 2, 3, 1).contiguous().to(dev), None, 1, 1, 0)
    finally:
        lib.ha2g_gemm_set_mode(__import__('ha2g_amd._lib', fromlist=['x']).DEFAULT_GEMM_MODE)
    assert relerr(c, a.double() @ b.double().t()) < 1e-5
    assert relerr(y.permute(0, 3, 1, 2), F.conv2d(x.double(), w.double(), padding=1)) < 1e-5


def test_three_piece_split_core_is_fp32_accurate():
    """The opt-in 3-piece forward matrix core (3 bf16 pieces per operand, 6 MFMAs, fp32 accumulate) is as accurate as the fp32 MFMA
    chain: against float64, its error is within 1.5x of the exact-fp32 mode's on a GEMM (ragged K) and a convolution."""
    import torch.nn.functional as F
    from ha2g_amd import ops, wav_engine as we
    from ha2g_amd._lib import lib
    dev = _dev()
    a, b = rnd((1000, 348), 51), rnd((300, 348), 52)
    x = rnd((2, 64, 20, 12), 53)
    w = rnd((96, 64, 3, 3), 54, 0.05)
    ref_c = a.double() @ b.double().t()
    ref_y = F.conv2d(x.double(), w.double(), padding=1)
    out = {}
    try:
        for mode in (0, 14):
            lib.ha2g_gemm_set_mode(mode)
            c = ops.gemm(a.to(dev), b.to(dev), transb=True)
            y = we.conv_fwd(x.permute(0, 2, 3, 1).contiguous().to(dev), w.permute(0, 2, 3, 1).contiguous().to(dev), None, 1, 1, 0)
            rms = lambda g, r: float(((g.double().cpu() - r) ** 2).mean().sqrt() / (r ** 2).mean().sqrt())
            out[mode] = (rms(c, ref_c), rms(y.permute(0, 3, 1, 2), ref_y))
    finally:
        lib.ha2g_gemm_set_mode(__import__('ha2g_amd._lib', fromlist=['x']).DEFAULT_GEMM_MODE)
    assert out[14][0] <= 1.5 * out[0][0] + 1e-8 and out[14][0] < 5e-7, out
    assert out[14][1] <= 1.5 * out[0][1] + 1e-8 and out[14][1] < 5e-7, out


def test_log_mel_frontend():
    """On-GPU log-mel (ha2g_logmel_f32) vs the NumPy restatement of librosa's defaults (oracle/logmel_oracle.py; parity
    unpinned: librosa absent).  Clips of the length the loader uses (34 frames at 15 fps = 36 267 samples -> 71 columns,
    cropped to 70), tones + noise at different levels, both padding modes; tolerance 0.05 dB before float16 rounding
    (float16 spacing at -64..-80 dB is 0.0625), at most one float16 ulp after."""
    import numpy as np
    from ha2g_amd import audio_frontend as fe
    from oracle import logmel_oracle as L
    dev = _dev()
    n = 36267
    t = np.arange(n) / 16000.0
    r = np.random.Generator(np.random.PCG64(11))
    clips = np.stack([0.3 * np.sin(2 * np.pi * 440 * t) + 0.02 * r.standard_normal(n),
                      0.05 * np.sin(2 * np.pi * 3000 * t + 1.0) * np.exp(-3 * t) + 1e-3 * r.standard_normal(n),
                      0.5 * r.standard_normal(n) * (t > 1.0)]).astype(np.float32)
    for mode in ('reflect', 'constant'):
        got = fe.batch_log_mel(torch.from_numpy(clips).to(dev), pad_mode=mode, f16=False).cpu().numpy()
        got16 = fe.batch_log_mel(torch.from_numpy(clips).to(dev), pad_mode=mode, f16=True).cpu().numpy()
        assert got.shape == (3, 128, 71)
        for b in range(3):
            ref = L.extract_melspectrogram(clips[b].astype(np.float64), pad_mode=mode, f16=False)
            assert np.abs(got[b] - ref).max() < 0.05, (mode, b, float(np.abs(got[b] - ref).max()))
            ref16 = ref.astype(np.float16).astype(np.float32)
            assert np.abs(got16[b] - ref16).max() <= 0.0626, (mode, b)
            assert float(got[b].max()) == 0.0 and float(got[b].min()) >= -80.0
    one = fe.extract_melspectrogram(clips[0])
    assert one.dtype == torch.float16 and tuple(one.shape) == (128, 71)


@pytest.mark.parametrize('shape', [(3, 20, 14), (2, 128, 70), (1, 16, 16), (2, 9, 37), (3, 17, 66), (1, 2, 78), (5, 12, 31)])
def test_direct_conv3x3_c32(shape):
    """The direct LDS-patch kernel for the 32->32 channel 3x3 convolutions (layer1 of the audio tower; conv_c32.hip), forward
    (+ReLU) and data gradient (accumulating), against float64 torch and against the implicit-GEMM path it replaces.  Shapes:
    ragged last tile, the real 128x70 map, exactly one tile, a width that makes tiles straddle many rows; round 4 (anti-phase kernel, two groups of
    four waves on alternating 128-pixel tiles): an odd number of tiles (one group idles in the last round), the widest and a narrow map it serves."""
    import torch.nn.functional as F
    from ha2g_amd import wav_engine as we
    from ha2g_amd._lib import lib
    dev = _dev()
    N, H, W = shape
    x = rnd((N, 32, H, W), 61)
    w = rnd((32, 32, 3, 3), 62, 0.08)
    dy = rnd((N, 32, H, W), 63)
    base = rnd((N, 32, H, W), 64)
    xg, dyg, wg = (t.permute(0, 2, 3, 1).contiguous().to(dev) for t in (x, dy, w))
    ref_y = F.relu(F.conv2d(x.double(), w.double(), padding=1))
    ref_dx = base.double() + F.conv_transpose2d(dy.double(), w.double(), padding=1)
    outs = {}
    try:
        # 3: fp32 direct kernel for forward and data gradient; 0: forward implicit GEMM, data gradient on the split-bf16 direct kernel
        # (the default is 1: direct forward + split-bf16 direct data gradient); 4: everything on the implicit GEMM
        # 67 = 3 + bit 6: the fp32 direct forward kernel also in the default (three-piece) arithmetic mode, where 3 sends the forward to the three-piece
        # anti-phase kernel (conv3x3_c32pp_kernel) like the data gradient
        for mode in (3, 67, 0, 4):
            lib.ha2g_conv_debug_direct_c32(mode)
            y = we.conv_fwd(xg, wg, None, 1, 1, we.ACT_RELU)
            acc = base.permute(0, 2, 3, 1).contiguous().to(dev)
            dx = we.conv_dgrad(dyg, wg, (N, H, W, 32), 1, 1, out=acc, beta=1.0)
            outs[mode] = (y.permute(0, 3, 1, 2), dx.permute(0, 3, 1, 2))
    finally:
        lib.ha2g_conv_debug_direct_c32(__import__('ha2g_amd._lib', fromlist=['x']).DEFAULT_DIRECT_C32)
    for mode in (3, 67, 0, 4):
        assert relerr(outs[mode][0], ref_y) < 2e-6, mode
        # split-bf16 data gradients: ~4e-6 rms; the direct kernels need W + 2 > 64, narrower maps fall back to the split GEMM
        assert relerr(outs[mode][1], ref_dx) < (2e-6 if (mode == 3 and W > 62) else 2e-5), mode
    # the direct forward kernel feeds every accumulator the implicit GEMM's MFMA chain (same k pairs, same order): bit-identical outputs
    assert torch.equal(outs[67][0], outs[4][0])


def test_plain_bf16_mode_is_bf16_accurate():
    """Mode bit 4 (bench --bf16, BASELINE config 5): GEMMs / convolutions with plain bf16 operands and fp32 accumulation are
    within bf16 rounding (2^-9 per operand -> ~3e-3 rms) of float64 for every loader mode; never enabled by default."""
    import torch.nn.functional as F
    from ha2g_amd import ops, wav_engine as we
    from ha2g_amd._lib import lib
    dev = _dev()
    a, b = rnd((1000, 600), 71), rnd((300, 600), 72)
    b2 = rnd((600, 300), 73)
    x = rnd((2, 64, 20, 12), 74)
    w = rnd((64, 64, 3, 3), 75, 0.05)
    dy = rnd((2, 64, 20, 12), 76)
    rms = lambda g, r: float(((g.double().cpu() - r) ** 2).mean().sqrt() / (r ** 2).mean().sqrt())
    lib.ha2g_gemm_set_mode(22)
    try:
        c_nt = ops.gemm(a.to(dev), b.to(dev), transb=True)
        c_nn = ops.gemm(a.to(dev), b2.to(dev))
        c_tn = ops.gemm(a.to(dev), a.to(dev)[:, :300].contiguous(), transa=True)
        xg, wg, dyg = (t.permute(0, 2, 3, 1).contiguous().to(dev) for t in (x, w, dy))
        y = we.conv_fwd(xg, wg, None, 1, 1, 0)
        dx = we.conv_dgrad(dyg, wg, (2, 20, 12, 64), 1, 1)
        dw = we.conv_wgrad(xg, dyg, wg, 1, 1)
    finally:
        lib.ha2g_gemm_set_mode(__import__('ha2g_amd._lib', fromlist=['x']).DEFAULT_GEMM_MODE)
    refs = [(c_nt, a.double() @ b.double().t()), (c_nn, a.double() @ b2.double()), (c_tn, a.double().t() @ a.double()[:, :300]),
            (y.permute(0, 3, 1, 2), F.conv2d(x.double(), w.double(), padding=1)),
            (dx.permute(0, 3, 1, 2), F.conv_transpose2d(dy.double(), w.double(), padding=1)),
            (dw, torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), padding=1))]
    for i, (got, ref) in enumerate(refs):
        e = rms(got, ref)
        assert 1e-4 < e < 6e-3, (i, e)          # bf16-level error: not fp32 (would be < 1e-6), not broken


@pytest.mark.parametrize('expr', [False, True])
@pytest.mark.parametrize('N', [4352, 8704])
def test_contrastive_full_size_vs_oracle(N, expr):
    """SoftmaxContrastiveLoss at the headline sizes -- N = B*34 = 4352 (B=128) and 8704 (B=256: the row-blocked branch
    above 80 MB of pair matrix) -- against the oracle's float64 loss (train_hierarchy.py:54-68 / expressive :107-121),
    back-propagated one 128-row block at a time on the host."""
    from ha2g_amd import ops
    from oracle import ha2g_oracle as O
    torch.set_num_threads(max(1, min(64, (torch.get_num_threads() or 8))))
    a, b = rnd((N, 32), 31, 1.0), rnd((N, 32), 32, 1.0)
    b[: N // 2] = a[: N // 2] + 0.3 * b[: N // 2]                   # correlated pairs: the diagonal logits matter
    a64, b64 = a.double().requires_grad_(True), b.double().requires_grad_(True)
    loss64 = 0.0
    for s in range(0, N, 128):
        l = O.contrastive_ce_rows(a64, b64, s, min(s + 128, N), expr)
        l.backward()
        loss64 += float(l)
    ag, bg = a.to(_dev()).requires_grad_(True), b.to(_dev()).requires_grad_(True)
    loss = ops.contrastive(ag, bg, expr)
    loss.backward()
    assert abs(float(loss) - loss64) <= 2e-5 * abs(loss64), (float(loss), loss64)
    assert relerr(ag.grad, a64.grad) < 1e-4 and relerr(bg.grad, b64.grad) < 1e-4, (relerr(ag.grad, a64.grad), relerr(bg.grad, b64.grad))


def test_dense_forward_split_leaves_the_audio_tower_bits_unchanged():
    """Mode bit 5 (3-piece forward split for dense GEMMs) must not touch the audio encoder: its forward arithmetic is frozen.
    Same outputs bit for bit with the bit on and off; a generator-sized GEMM does change (and stays fp32-accurate)."""
    from ha2g_amd import ops, procedural as proc
    from ha2g_amd._lib import DEFAULT_GEMM_MODE, lib
    from ha2g_amd.config import CASES
    from ha2g_testing import batch_for, build_modules
    dev = _dev()
    case = CASES['small']
    _, _, _, aud, _ = build_modules(case, dev)
    _, spec, _, vid = batch_for(case)
    a, b = rnd((4352, 600), 3).to(dev), rnd((900, 600), 4, 600 ** -0.5).to(dev)
    outs = {}
    try:
        for mode in (6, 38):
            lib.ha2g_gemm_set_mode(mode)
            with torch.no_grad():
                w, lo, mid, hi, blend = aud(spec.to(dev), vid.to(dev))
            outs[mode] = (torch.cat([lo.reshape(-1), mid.reshape(-1), hi.reshape(-1), w.reshape(-1)] + [t.reshape(-1) for t in blend]),
                          ops.gemm(a, b, transb=True))
    finally:
        lib.ha2g_gemm_set_mode(DEFAULT_GEMM_MODE)
    assert torch.equal(outs[6][0], outs[38][0])
    assert not torch.equal(outs[6][1], outs[38][1])
    ref = a.double().cpu() @ b.double().cpu().t()
    assert relerr(outs[38][1], ref) < 2e-6 and relerr(outs[6][1], ref) < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(4352, 600, 300, False, False), (1000, 300, 600, False, True), (136, 900, 600, False, True),
                                   (4352, 600, 900, False, False), (2176, 300, 150, False, False)])
def test_dense_tile_shape_does_not_change_bits(shape):
    """The tile shape of a dense GEMM (nine shapes, picked per problem by a fitted time model in round 2) is a pure speed choice: for a fixed
    split-K count -- still decided by round 1's rule -- every output element runs the same MFMA chain over k in the same order."""
    import ctypes
    from ha2g_amd import ops
    from ha2g_amd._lib import lib
    M, N, K, ta, tb = shape
    g = torch.Generator(device='cuda:0').manual_seed(M + N)
    a = torch.randn((K, M) if ta else (M, K), device='cuda:0', generator=g)
    b = torch.randn((N, K) if tb else (K, N), device='cuda:0', generator=g)
    try:
        lib.ha2g_gemm_debug_tile(-2, 0)                                    # round 1's tile rule
        ref = ops.gemm(a, b, transa=ta, transb=tb).clone()
        # three-piece mode, data-gradient shape (n-contiguous B): the [m][k] planes of tiles 0 / 5 / 6 exceed the 64 KB of static LDS and those tiles
        # run the exact fp32 MFMA instead (dispatch_tile never picks them there) -- a different arithmetic by design, not a tile effect
        skip = (0, 5, 6) if (lib.ha2g_gemm_bwd_pieces() == 3 and not ta and not tb) else ()
        if 0 in skip:
            lib.ha2g_gemm_debug_tile(2, 0)
            ref = ops.gemm(a, b, transa=ta, transb=tb).clone()
        for cfg in [-1] + list(range(9)):
            if cfg in skip:
                continue
            lib.ha2g_gemm_debug_tile(cfg, 0)
            assert torch.equal(ops.gemm(a, b, transa=ta, transb=tb), ref), cfg
    finally:
        lib.ha2g_gemm_debug_tile(-1, 0)
    ref64 = (a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double())
    assert float((ref.double() - ref64).abs().max() / ref64.abs().max()) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(300, 600, 4352), (900, 600, 4352), (150, 300, 4352), (27, 150, 4352), (64, 16, 128), (300, 300, 136),
                                   (129, 77, 1000), (16, 128, 7168)])
@pytest.mark.parametrize('acc', [False, True])
def test_wgrad_gemm_with_fused_bias_gradient(shape, acc):
    """ha2g_gemm_wgrad_bias_f32: the weight gradient is bit-identical to the plain GEMM's, the bias gradient equals the float64 column sums of dY
    (split-K and single-pass launches, 16-byte-aligned and ragged shapes, overwrite and accumulate)."""
    from ha2g_amd import ops
    M, N, K = shape
    g = torch.Generator(device='cuda:0').manual_seed(M * 7 + N)
    dy = torch.randn(K, M, device='cuda:0', generator=g) + 0.25
    x = torch.randn(K, N, device='cuda:0', generator=g)
    w0 = torch.randn(M, N, device='cuda:0', generator=g) if acc else torch.zeros(M, N, device='cuda:0')
    b0 = torch.randn(M, device='cuda:0', generator=g) if acc else torch.full((M,), float('nan'), device='cuda:0')
    ref_w = ops.gemm(dy, x, transa=True, out=w0.clone(), beta=1.0 if acc else 0.0)
    got_w, got_b = w0.clone(), b0.clone()
    ops.gemm(dy, x, transa=True, out=got_w, beta=1.0 if acc else 0.0, colsum_out=got_b, colsum_beta=1.0 if acc else 0.0)
    assert torch.equal(got_w, ref_w)
    ref_b = dy.double().sum(0) + (b0.double() if acc else 0.0)
    scale = float(dy.double().abs().sum(0).max())
    assert float((got_b.double() - ref_b).abs().max()) <= 2e-6 * scale


@pytest.mark.gpu
def test_linear_and_conv1d_bias_gradients_fused_equal_unfused():
    from ha2g_amd import ops
    g = torch.Generator(device='cuda:0').manual_seed(11)
    x = torch.randn(64, 34, 48, device='cuda:0', generator=g, requires_grad=True)
    w = torch.randn(40, 48, device='cuda:0', generator=g, requires_grad=True)
    b = torch.randn(40, device='cuda:0', generator=g, requires_grad=True)
    wc = torch.randn(40, 48, 2, device='cuda:0', generator=g, requires_grad=True)
    res = {}
    for fuse in (True, False):
        ops.FUSE_BIAS_GRAD = fuse
        try:
            y = ops.linear(x, w, b, ops.ACT_LEAKY) .sum() + (ops.conv1d_tm(x, wc, b, 2, 2, 34, ops.ACT_RELU) ** 2).sum()
            res[fuse] = torch.autograd.grad(y, (x, w, b, wc))
        finally:
            ops.FUSE_BIAS_GRAD = True
    for a_, b_ in zip(res[True], res[False]):
        assert float((a_ - b_).abs().max()) <= 2e-6 * float(b_.abs().max())
    assert torch.equal(res[True][1], res[False][1]) and torch.equal(res[True][3], res[False][3])


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(3, 4352, 300, 600, False, True), (3, 4352, 600, 300, False, False), (3, 300, 600, 4352, True, False),
                                  (6, 136, 32, 300, False, True), (2, 129, 77, 1000, True, False), (3, 8704, 300, 600, False, True),
                                  (1, 300, 300, 4352, True, False)])
def test_grouped_gemm_matches_per_group_launches(case):
    """ha2g_gemm_grouped_f32: G problems of one shape in one launch == G plain launches (forward with bias + activation, data-gradient and
    weight-gradient shapes with the fused bias gradient; stacked and separate operands; accumulate)."""
    from ha2g_amd import ops
    G, M, N, K, ta, tb = case
    gen = torch.Generator(device='cuda:0').manual_seed(G * 1000 + M)
    a = torch.randn(G, *((K, M) if ta else (M, K)), device='cuda:0', generator=gen)
    b = [torch.randn((N, K) if tb else (K, N), device='cuda:0', generator=gen) for _ in range(G)]         # separate tensors (weights)
    c0 = torch.randn(G, M, N, device='cuda:0', generator=gen)
    wgrad = ta and not tb
    bias = None if wgrad else torch.randn(G, N, device='cuda:0', generator=gen)
    cs0 = torch.randn(G, M, device='cuda:0', generator=gen) if wgrad else None
    act = ops.ACT_NONE if wgrad else ops.ACT_RELU
    got, got_cs = c0.clone(), (cs0.clone() if wgrad else None)
    ops.gemm_grouped(a, b, transa=ta, transb=tb, out=got, beta=1.0, bias=bias, act=act, colsum_out=got_cs, colsum_beta=1.0)
    for g in range(G):
        ref = a[g].double().t() @ b[g].double() if ta else a[g].double() @ (b[g].double().t() if tb else b[g].double())
        ref = ref + c0[g].double() + (bias[g].double() if bias is not None else 0.0)
        if act == ops.ACT_RELU:
            ref = ref.clamp_min(0)
        scale = float(ref.abs().max())
        assert float((got[g].double() - ref).abs().max()) <= 3e-5 * scale, g                 # split-bf16 class on the gradient shapes
        if wgrad:
            refb = a[g].double().sum(0) + cs0[g].double()
            assert float((got_cs[g].double() - refb).abs().max()) <= 2e-6 * float(a[g].double().abs().sum(0).max()), g
    if G == 1:                                                                             # degenerate group count == the plain launch, bit for bit
        one, one_cs = c0[0].clone(), (cs0[0].clone() if wgrad else None)
        pg = ops.PLANE_GEMM
        ops.PLANE_GEMM = False                          # the plain launch of the SAME kernel (large products otherwise take the plane GEMM)
        try:
            ops.gemm(a[0], b[0], transa=ta, transb=tb, out=one, beta=1.0, bias=None if bias is None else bias[0], act=act,
                     **(dict(colsum_out=one_cs, colsum_beta=1.0) if wgrad else {}))
        finally:
            ops.PLANE_GEMM = pg
        assert torch.equal(one, got[0])


@pytest.mark.gpu
def test_plane_staged_weight_gradients_are_bit_identical_to_the_packed_word_kernel():
    """SPLIT = 3 (bf16 hi / lo planes + ds_read_b64_tr_b16 transpose reads, the default for weight-gradient shapes) against SPLIT = 1 (packed
    {hi|lo} words rebuilt into fragments with VALU ops): same split values, same MFMA order -> torch.equal; dense (every tile shape the rule
    can pick, with the fused bias gradient), grouped, and the implicit-GEMM convolution weight gradients of every channel width."""
    from ha2g_amd import ops, wav_engine as we
    from ha2g_amd._lib import lib
    g = torch.Generator(device='cuda:0').manual_seed(77)

    def both(fn):
        out = []
        for planes in (0, 1):
            lib.ha2g_conv_debug_cfg(30000 + planes)
            out.append(fn())
        lib.ha2g_conv_debug_cfg(30001)
        return out
    try:
        for (M, N, K) in [(300, 600, 4352), (900, 600, 4352), (150, 300, 4352), (64, 16, 1000), (32, 288, 2048), (600, 300, 13056), (27, 150, 4352)]:
            dy, x = torch.randn(K, M, device='cuda:0', generator=g), torch.randn(K, N, device='cuda:0', generator=g)

            def run():
                w, b = torch.zeros(M, N, device='cuda:0'), torch.zeros(M, device='cuda:0')
                if M % 4 == 0:
                    ops.gemm(dy, x, transa=True, out=w, colsum_out=b)
                else:
                    ops.gemm(dy, x, transa=True, out=w)
                return w, b
            (w0, b0), (w1, b1) = both(run)
            assert torch.equal(w0, w1) and torch.equal(b0, b1), (M, N, K)
            ref = dy.double().t() @ x.double()
            assert float((w1.double() - ref).abs().max() / ref.abs().max()) < 2e-5
        a = torch.randn(3, 4352, 300, device='cuda:0', generator=g); b = torch.randn(3, 4352, 600, device='cuda:0', generator=g)
        o0, o1 = both(lambda: ops.gemm_grouped(a, b, transa=True))
        assert torch.equal(o0, o1)
        # dense data gradients dX = dY W (k-contiguous A in [m][k] planes with plain 16-byte reads, n-contiguous B through the transpose reads)
        for (M, N, K) in [(4352, 600, 900), (4352, 600, 300), (7168, 128, 192), (1000, 300, 152), (136, 900, 600), (4352, 32, 300)]:
            dy, w = torch.randn(M, K, device='cuda:0', generator=g), torch.randn(K, N, device='cuda:0', generator=g)
            d0, d1 = both(lambda: ops.gemm(dy, w))
            assert torch.equal(d0, d1), (M, N, K)
            ref = dy.double() @ w.double()
            assert float((d1.double() - ref).abs().max() / ref.abs().max()) < 2e-5
        w3 = [torch.randn(300, 600, device='cuda:0', generator=g) for _ in range(3)]
        o0, o1 = both(lambda: ops.gemm_grouped(a, w3))
        assert torch.equal(o0, o1)
        for (N, H, W, Cin, Cout, stride) in [(4, 128, 70, 32, 32, 1), (4, 64, 35, 64, 64, 1), (4, 32, 18, 128, 128, 1), (4, 16, 9, 256, 256, 1),
                                             (4, 128, 70, 32, 64, 2)]:
            x = torch.randn(N, H, W, Cin, device='cuda:0', generator=g)
            OH, OW = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
            dy = torch.randn(N, OH, OW, Cout, device='cuda:0', generator=g)
            w0 = torch.empty(Cout, 3, 3, Cin, device='cuda:0')
            d0, d1 = both(lambda: we.conv_wgrad(x, dy, w0, stride, 1).clone())
            assert torch.equal(d0, d1), (Cin, Cout, stride)
    finally:
        lib.ha2g_conv_debug_cfg(30001)


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(4, 128, 70), (3, 13, 70), (2, 16, 64), (1, 9, 35), (5, 64, 35)])
def test_direct_c32_weight_gradient(shape):
    """conv_c32.hip's direct 32-channel weight gradient (patch + dy strip as bf16 planes, transpose-read fragments, nine taps share one dy
    fragment) against float64 autograd and against the implicit GEMM it replaces; ragged last tile, tiles that straddle many image rows, a map
    too small for the kernel (falls back), accumulation into an existing gradient."""
    import torch.nn.functional as F
    from ha2g_amd import wav_engine as we
    from ha2g_amd._lib import lib
    N, H, W = shape
    g = torch.Generator(device='cuda:0').manual_seed(H * W)
    x = torch.randn(N, H, W, 32, device='cuda:0', generator=g)
    dy = torch.randn(N, H, W, 32, device='cuda:0', generator=g)
    w0 = torch.empty(32, 3, 3, 32, device='cuda:0')
    xd = x.permute(0, 3, 1, 2).double()
    wd = torch.zeros(32, 32, 3, 3, dtype=torch.float64, device='cuda:0', requires_grad=True)
    (F.conv2d(xd, wd, padding=1) * dy.permute(0, 3, 1, 2).double()).sum().backward()
    ref = wd.grad
    outs = {}
    try:
        for direct in (1, 0):
            lib.ha2g_conv_debug_cfg(40000 + direct)
            outs[direct] = we.conv_wgrad(x, dy, w0, 1, 1).clone()
    finally:
        lib.ha2g_conv_debug_cfg(40001)
    scale = float(ref.abs().max())
    for direct in (1, 0):
        assert float((outs[direct].double() - ref).abs().max()) <= 2e-5 * scale, direct
    acc = torch.randn(32, 32, 3, 3, device='cuda:0', generator=g).contiguous(memory_format=torch.channels_last)
    base = acc.clone()
    we.conv_wgrad(x, dy, w0, 1, 1, into=acc)
    assert float((acc.double() - (base.double() + ref)).abs().max()) <= 2e-5 * scale


@pytest.mark.gpu
@pytest.mark.parametrize('case', [  # (G, M, N, K, transa, transb, bias / csum, act, beta)
    (1, 900, 600, 4352, True, False, True, 0, 1.0),       # GRU dW_ih: 7 k slices, fused bias gradient, accumulate into .grad
    (2, 900, 600, 4352, True, False, True, 0, 1.0),       # ... both directions grouped
    (3, 300, 600, 4352, True, False, True, 0, 0.0),       # TCN dW grouped: 12 slices
    (1, 150, 300, 4352, True, False, False, 0, 0.0),      # 34 slices over 6 tiles ("wide" reduce before)
    (1, 129, 77, 1000, True, False, True, 0, 1.0),        # ragged: scalar paths of the in-kernel reduction
    (1, 96, 40, 2048, False, True, True, 2, 1.0),         # forward shape with bias + leaky-relu + beta through the slabs
    (1, 64, 200, 1536, False, False, False, 1, 0.0),      # data-gradient shape, relu
    (1, 4352, 600, 1800, False, False, False, 0, 0.0),    # GRU dX: the plane GEMM (three k slices of the q kernel, reduced by the tile's last arriver)
    (1, 4352, 600, 1800, False, False, False, 0, 1.0),
    (1, 1024, 384, 2560, False, True, True, 2, 1.0),      # plane GEMM, transb, bias + leaky-relu + beta
])
def test_split_k_reduced_in_the_kernel_by_the_last_arriver(case):
    """Round 6 (built, measured, NOT the default: profiles/r06_splitk_inkernel.txt): with ha2g_splitk_in_kernel(1) a split-K launch whose
    stream has arrival tickets registered (ops.workspace does that) adds its slabs IN THE KERNEL -- the last k slice of an output tile to arrive sums them
    in slice order, in double -- instead of leaving them to splitk_reduce[_wide]_kernel.  Same arithmetic, element for element: the result is BIT-IDENTICAL
    to the two-launch form (the default), equal to float64 at the split-product bound, bitwise stable over 25 repetitions (arrival order varies, summation
    order does not; the tickets re-zero themselves), with the fused bias gradient reduced alike."""
    from ha2g_amd import ops
    from ha2g_amd._lib import lib
    G, M, N, K, ta, tb, extra, act, beta = case
    gen = torch.Generator(device='cuda:0').manual_seed(M * 31 + N * 7 + K + G)
    A = [torch.randn((K, M) if ta else (M, K), device='cuda:0', generator=gen) for _ in range(G)]
    Bm = [torch.randn((N, K) if tb else (K, N), device='cuda:0', generator=gen) for _ in range(G)]
    C0 = [torch.randn(M, N, device='cuda:0', generator=gen) for _ in range(G)]
    wgrad = ta and not tb
    bias = [torch.randn(N, device='cuda:0', generator=gen) for _ in range(G)] if (extra and not wgrad) else None
    cs0 = [torch.randn(M, device='cuda:0', generator=gen) for _ in range(G)] if (extra and wgrad) else None

    def run():
        out = [c.clone() for c in C0]
        cs = [c.clone() for c in cs0] if cs0 is not None else None
        kw = dict(transa=ta, transb=tb, beta=beta, act=act)
        if G == 1:
            ops.gemm(A[0], Bm[0], out=out[0], bias=None if bias is None else bias[0], **kw,
                     **(dict(colsum_out=cs[0], colsum_beta=1.0) if cs is not None else {}))
        else:
            ops.gemm_grouped(A, Bm, out=out, bias=bias, colsum_out=cs, colsum_beta=1.0, **kw)
        torch.cuda.synchronize()
        return out, cs
    ops.workspace(torch.device('cuda:0'))                     # registers the tickets of this stream
    ref, ref_cs = run()                                       # default: the two-launch form
    lib.ha2g_splitk_in_kernel(1)
    try:
        got, got_cs = run()
        for _ in range(25):
            again, again_cs = run()
            assert all(torch.equal(a, b) for a, b in zip(again, got))
            assert cs0 is None or all(torch.equal(a, b) for a, b in zip(again_cs, got_cs))
    finally:
        lib.ha2g_splitk_in_kernel(0)
    plane = G == 1 and 2.0 * M * N * K >= ops.PLANE_GEMM_MIN_FLOP and min(M, N) >= 128      # the plane GEMM also re-plans its k slices: fp32-level agreement
    for g in range(G):
        if plane:
            assert float((got[g] - ref[g]).abs().max()) <= 2e-6 * float(ref[g].abs().max())
        else:
            assert torch.equal(got[g], ref[g]), (g, float((got[g] - ref[g]).abs().max()))
        if cs0 is not None:
            assert torch.equal(got_cs[g], ref_cs[g])
        r64 = A[g].double().t() @ Bm[g].double() if ta else A[g].double() @ (Bm[g].double().t() if tb else Bm[g].double())
        r64 = r64 + (bias[g].double() if bias is not None else 0.0) + beta * C0[g].double()
        if act == 1:
            r64 = r64.clamp_min(0)
        elif act == 2:
            r64 = torch.where(r64 > 0, r64, 0.01 * r64)
        assert float((got[g].double() - r64).abs().max()) <= 3e-5 * float(r64.abs().max())
    tk = ops._tickets[('cuda', 0, ops._raw_stream(0))]
    assert int(tk.abs().sum()) == 0                            # every ticket is back at zero


@pytest.mark.gpu
@pytest.mark.parametrize('N,C,R', [(128, 32, 4), (128, 64, 8), (128, 128, 16), (128, 256, 32), (5, 48, 6)])
def test_se_mlp_parameter_gradients_in_one_launch(N, C, R):
    """ha2g_se_mlp_wgrad_f32 (round 6): dW / db of the SE excitation MLP's two Linear layers (model/ResNetBlocks.py:84-89), accumulated into existing
    buffers, against float64; and through GradSink.gse == the two generic weight-gradient GEMMs it replaces at fp32 level."""
    from ha2g_amd._lib import check, lib
    from ha2g_amd.ops import _stream
    g = torch.Generator(device='cuda:0').manual_seed(N * 1000 + C)
    dsc, pooled = torch.randn(N, C, device='cuda:0', generator=g), torch.randn(N, C, device='cuda:0', generator=g)
    h1, dh1 = torch.randn(N, R, device='cuda:0', generator=g).clamp_min(0), torch.randn(N, R, device='cuda:0', generator=g)
    init = [torch.randn(C, R, device='cuda:0', generator=g), torch.randn(C, device='cuda:0', generator=g),
            torch.randn(R, C, device='cuda:0', generator=g), torch.randn(R, device='cuda:0', generator=g)]
    out = [t.clone() for t in init]
    check(lib.ha2g_se_mlp_wgrad_f32(dsc.data_ptr(), h1.data_ptr(), dh1.data_ptr(), pooled.data_ptr(), out[0].data_ptr(), out[1].data_ptr(),
                                    out[2].data_ptr(), out[3].data_ptr(), N, C, R, _stream()))
    ref = [init[0].double() + dsc.double().t() @ h1.double(), init[1].double() + dsc.double().sum(0),
           init[2].double() + dh1.double().t() @ pooled.double(), init[3].double() + dh1.double().sum(0)]
    for o, r in zip(out, ref):
        assert float((o.double() - r).abs().max()) <= 2e-6 * max(float(r.abs().max()), 1.0) * (N ** 0.5)


@pytest.mark.gpu
def test_se_mlp_parameter_gradients_of_sixteen_blocks_in_one_launch_are_bit_identical():
    """ha2g_se_mlp_wgrad_multi_f32 (round 6): the tower's sixteen SE layers (widths 32 / 64 / 128 / 256, reduction 8) in one launch == sixteen
    ha2g_se_mlp_wgrad_f32 launches, bit for bit; through GradSink.gse / flush_se as the step calls it."""
    import numpy as np
    from ha2g_amd._lib import check, lib
    from ha2g_amd.ops import _stream
    N = 24
    g = torch.Generator(device='cuda:0').manual_seed(5)
    widths = [32] * 3 + [64] * 4 + [128] * 6 + [256] * 3
    jobs, one, multi = [], [], []
    for C in widths:
        R = C // 8
        dsc, pooled = torch.randn(N, C, device='cuda:0', generator=g), torch.randn(N, C, device='cuda:0', generator=g)
        h1, dh1 = torch.randn(N, R, device='cuda:0', generator=g).clamp_min(0), torch.randn(N, R, device='cuda:0', generator=g)
        init = [torch.randn(C, R, device='cuda:0', generator=g), torch.randn(C, device='cuda:0', generator=g),
                torch.randn(R, C, device='cuda:0', generator=g), torch.randn(R, device='cuda:0', generator=g)]
        jobs.append((dsc, h1, dh1, pooled))
        one.append([t.clone() for t in init])
        multi.append([t.clone() for t in init])
    for (dsc, h1, dh1, pooled), o in zip(jobs, one):
        check(lib.ha2g_se_mlp_wgrad_f32(dsc.data_ptr(), h1.data_ptr(), dh1.data_ptr(), pooled.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(),
                                        o[3].data_ptr(), N, dsc.shape[1], h1.shape[1], _stream()))
    n = len(jobs)
    ptr = np.empty((8, n), np.int64)
    for i, (job, o) in enumerate(zip(jobs, multi)):
        ptr[:, i] = [t.data_ptr() for t in job] + [t.data_ptr() for t in o]
    Cs, Rs = np.array(widths, np.int32), np.array([c // 8 for c in widths], np.int32)
    check(lib.ha2g_se_mlp_wgrad_multi_f32(n, *(ptr[k].ctypes.data for k in range(8)), Cs.ctypes.data, Rs.ctypes.data, N, _stream()))
    for a, b in zip(one, multi):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert lib.ha2g_se_mlp_wgrad_multi_f32(17, *(ptr[k].ctypes.data for k in range(8)), Cs.ctypes.data, Rs.ctypes.data, N, _stream()) != 0      # > 16 jobs: refused


@pytest.mark.gpu
@pytest.mark.parametrize('geom', [(4, 128, 70), (3, 64, 35), (2, 40, 37), (5, 128, 70)])
def test_c32_weight_gradient_prefetching_form_is_bit_identical(geom):
    """Round 6: the 32-channel three-piece weight gradient (layer 1 of the tower, side queue) loads the NEXT tile's patch and dy strip into registers in front
    of this tile's MFMA phase (ha2g_conv_c32_wgrad_prefetch(1)) -- same values, same summation order: BIT-IDENTICAL to the first form; mode 2 also
    uses 256-pixel tiles where the LDS holds them (fewer halo bytes per pixel; another, fixed tile order).  Both equal float64 at the three-piece bound."""
    from ha2g_amd import wav_engine as we
    from ha2g_amd._lib import lib
    N, H, W = geom
    g = torch.Generator(device='cuda:0').manual_seed(N * 100 + W)
    x = torch.randn(N, H, W, 32, device='cuda:0', generator=g)
    dy = torch.randn(N, H, W, 32, device='cuda:0', generator=g)
    w = torch.randn(32, 3, 3, 32, device='cuda:0', generator=g)
    out = {}
    for pf in (2, 1, 0):                                      # 1 = the default; 2 = + 256-pixel tiles where the LDS holds them (another tile order; no gain in the step)
        lib.ha2g_conv_c32_wgrad_prefetch(pf)
        try:
            out[pf] = we.conv_wgrad(x, dy, w, 1, 1).clone()
            again = we.conv_wgrad(x, dy, w, 1, 1)
            assert torch.equal(again, out[pf])
            torch.cuda.synchronize()
        finally:
            lib.ha2g_conv_c32_wgrad_prefetch(1)
    assert torch.equal(out[1], out[0])
    ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double(), (32, 32, 3, 3), dy.permute(0, 3, 1, 2).double(), padding=1)     # [co][ci][kh][kw]
    assert out[1].shape == ref.shape
    for pf in (2, 1):
        assert float((out[pf].double() - ref).abs().max()) <= 3e-6 * float(ref.abs().max()), pf

"""The fp32-class backward of round 4 (ha2g_gemm_set_mode bit 6, the default): every split product runs on THREE bf16 pieces per operand and six
MFMAs, x = p0 + p1 + p2 holding all 24 mantissa bits, products down to 2^-24 kept.  Each kernel family is held against float64 at the accuracy
of the exact fp32 MFMA chain (and against that chain itself), on the shapes the train step uses.  (Replaces autograd's fp32 loss.backward(),
train_eval/train_hierarchy.py:264; conv2d / linear backward of model/ResNetBlocks.py:24-29, model/hierarchy_net.py:87-93.)"""
import pytest
import torch

from ha2g_amd import ops, wav_engine as we
from ha2g_amd._lib import DEFAULT_GEMM_MODE, lib

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(autouse=True)
def three_piece_mode():
    lib.ha2g_gemm_set_mode(70)
    yield
    lib.ha2g_gemm_set_mode(DEFAULT_GEMM_MODE)


def test_default_mode_is_the_three_piece_backward():
    assert DEFAULT_GEMM_MODE & 64 and lib.ha2g_gemm_bwd_pieces() == 3 and ops.pieces() == 3
    lib.ha2g_gemm_set_mode(6)
    assert lib.ha2g_gemm_bwd_pieces() == 2
    lib.ha2g_gemm_set_mode(0)
    assert lib.ha2g_gemm_bwd_pieces() == 0


def test_three_pieces_hold_the_whole_fp32_mantissa():
    torch.manual_seed(0)
    x = torch.randn(1 << 16, device=DEV) * torch.logspace(-6, 6, 1 << 16, device=DEV)
    pl = ops.to_planes(x)
    assert pl.shape[0] == 3
    assert torch.equal(pl[0], x.bfloat16())
    r1 = x - pl[0].float()
    assert torch.equal(pl[1], r1.bfloat16())
    assert torch.equal(pl[2], (r1 - pl[1].float()).bfloat16())
    assert torch.equal(pl[0].float() + pl[1].float() + pl[2].float(), x)          # exact: nothing of the fp32 value is lost


def _rel(a, ref):
    return float((a.double() - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize('relu_mask', [False, True])
def test_bn_bwd_writes_three_piece_planes(relu_mask):
    torch.manual_seed(1)
    rows, C = 4 * 32 * 18, 128
    x = torch.relu(torch.randn(rows, C, device=DEV) + 0.2) if relu_mask else torch.randn(rows, C, device=DEV)
    dy = torch.randn(rows, C, device=DEV)
    mean, invstd = ops.bn_stats(x, None, None, 0.1, 1e-5)
    gamma = torch.rand(C, device=DEV) + 0.5
    dx0, dg0, db0 = ops.bn_bwd(dy, x, mean, invstd, gamma, relu_mask=relu_mask)
    dx1, dg1, db1, pl = ops.bn_bwd(dy, x, mean, invstd, gamma, relu_mask=relu_mask, planes=True)
    assert pl.shape[0] == 3 and torch.equal(dx0, dx1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
    assert torch.equal(pl, ops.to_planes(dx0))
    assert torch.equal(pl[0].float() + pl[1].float() + pl[2].float(), dx0)
    _, _, _, pl2 = ops.bn_bwd(dy, x, mean, invstd, gamma, need_dx=False, relu_mask=relu_mask, planes=True)
    assert torch.equal(pl2, pl)


@pytest.mark.parametrize('tile', [0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize('B,H,W,C', [(4, 64, 35, 64), (3, 32, 18, 128), (5, 16, 9, 256), (1, 7, 5, 64), (128, 32, 18, 128)])
def test_dgrad_three_piece_planes_are_fp32_class(B, H, W, C, tile):
    """3x3 stride-1 data gradients of trunk layers 2-4 on the DMA-staged plane kernel with three pieces: as close to float64 as the exact fp32
    MFMA kernel (mode 0) is -- 20x closer than the two-piece product -- for every tile shape / kernel form (1-3: 32x32 tiles, 4: eight-wave ping-pong,
    5: the quantisation-free 16x16 ping-pong kernel forced, 6: that kernel off, 0: the default choice), beta = 0 and beta = 1."""
    torch.manual_seed(2)
    dy = torch.randn(B, H, W, C, device=DEV)
    w = torch.randn(C, 3, 3, C, device=DEV) * 0.05
    assert we.dgrad_planes_ok(w, 1, 1)
    x64 = torch.nn.functional.conv_transpose2d(dy.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    lib.ha2g_conv_planes_tile3(tile)
    try:
        got = we.conv_dgrad_planes(ops.to_planes(dy), w, (B, H, W, C), 1, 1)
        base = torch.randn(B, H, W, C, device=DEV)
        got1 = we.conv_dgrad_planes(ops.to_planes(dy), w, (B, H, W, C), 1, 1, out=base.clone(), beta=1.0)
    finally:
        lib.ha2g_conv_planes_tile3(0)
    lib.ha2g_gemm_set_mode(0)
    exact = we.conv_dgrad(dy, w, (B, H, W, C), 1, 1)
    lib.ha2g_gemm_set_mode(70)
    e3, e32 = _rel(got, x64), _rel(exact, x64)
    assert e3 < 3e-6 and e3 < 1.5 * e32 + 2e-7, (e3, e32)          # K = 9 C up to 2304 terms: the fp32 chain itself sits at 1-2e-6 here
    assert _rel(got1 - base, x64) < 5e-6


@pytest.mark.parametrize('tile', [0, 4, 7])      # 0 = default (q kernel, the parity classes over grid.z), 4 / 7 = the 32x32 kernels
@pytest.mark.parametrize('B,H,W,Cin,Cout,k', [(4, 128, 70, 32, 64, 3), (3, 64, 35, 64, 128, 3), (5, 32, 18, 128, 256, 3), (2, 9, 7, 64, 64, 3),
                                              (4, 128, 70, 32, 64, 1), (3, 64, 35, 64, 128, 1), (4, 32, 18, 128, 256, 1)])
def test_stride2_dgrad_three_piece_planes_are_fp32_class(B, H, W, Cin, Cout, k, tile):
    torch.manual_seed(5)
    pad = 1 if k == 3 else 0
    OH, OW = (H + 2 * pad - k) // 2 + 1, (W + 2 * pad - k) // 2 + 1
    dy = torch.randn(B, OH, OW, Cout, device=DEV)
    w = torch.randn(Cout, k, k, Cin, device=DEV) * 0.05
    assert we.dgrad_planes_ok(w, 2, pad)
    lib.ha2g_conv_planes_tile3(tile)
    try:
        got = we.conv_dgrad_planes(ops.to_planes(dy), w, (B, H, W, Cin), 2, pad)
    finally:
        lib.ha2g_conv_planes_tile3(0)
    x64 = torch.nn.functional.conv_transpose2d(dy.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), stride=2, padding=pad,
                                               output_padding=(H - ((OH - 1) * 2 - 2 * pad + k), W - ((OW - 1) * 2 - 2 * pad + k))).permute(0, 2, 3, 1)
    assert _rel(got, x64) < 3e-6


@pytest.mark.parametrize('H,W,Cin,Cout,k,stride,pad', [(64, 35, 64, 64, 3, 1, 1), (32, 18, 128, 128, 3, 1, 1), (16, 9, 256, 256, 3, 1, 1), (128, 70, 32, 32, 3, 1, 1),
                                                       (128, 70, 32, 64, 3, 2, 1), (64, 35, 64, 128, 1, 2, 0), (64, 36, 32, 32, 3, 1, 0), (63, 34, 64, 64, 2, 1, 0)])
def test_conv_gradients_of_every_trunk_and_tap_geometry_are_fp32_class(H, W, Cin, Cout, k, stride, pad):
    """conv_dgrad / conv_wgrad as the tower's backward calls them (whatever kernel the library dispatches in the default mode: plane kernel,
    three-piece implicit GEMM, exact fp32 direct kernel) against float64 autograd."""
    torch.manual_seed(6)
    B = 3
    x = torch.randn(B, H, W, Cin, device=DEV)
    w = torch.randn(Cout, k, k, Cin, device=DEV) * 0.05
    xx = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    ww = w.double().permute(0, 3, 1, 2).requires_grad_(True)
    y = torch.nn.functional.conv2d(xx, ww, stride=stride, padding=pad)
    dy = torch.randn(y.shape, device=DEV, dtype=torch.float64)
    gx, gw = torch.autograd.grad(y, [xx, ww], dy)
    dyn = dy.float().permute(0, 2, 3, 1).contiguous()
    dx = we.conv_dgrad(dyn, w, x.shape, stride, pad)
    dw = we.conv_wgrad(x, dyn, w, stride, pad)
    assert _rel(dx, gx.permute(0, 2, 3, 1)) < 2e-6
    assert _rel(dw, gw) < 3e-6


@pytest.mark.parametrize('M,N,K', [(4352, 900, 600), (13056, 900, 600), (4352, 900, 108), (4352, 150, 300), (4352, 300, 600), (896, 300, 64), (4352, 32, 300)])
def test_dense_backward_products_are_fp32_class(M, N, K):
    """dW = dY^T X and dX = dY W of the GRU projections, the generator head, the TCN and the discriminator (ops.gemm as LinearFunction.backward
    calls it): three-piece plane tiles (gemm_kernel<.., SPLIT = 4>) or the exact fp32 MFMA where those do not serve the shape."""
    torch.manual_seed(7)
    dy = torch.randn(M, N, device=DEV)
    x = torch.randn(M, K, device=DEV)
    w = torch.randn(N, K, device=DEV) * 0.05
    dw = ops.gemm(dy, x, transa=True)
    dx = ops.gemm(dy, w)
    assert _rel(dw, dy.double().t() @ x.double()) < 2e-6
    assert _rel(dx, dy.double() @ w.double()) < 2e-6
    lib.ha2g_gemm_set_mode(6)
    dw2 = ops.gemm(dy, x, transa=True)
    lib.ha2g_gemm_set_mode(70)
    e2 = _rel(dw2, dy.double().t() @ x.double())
    if e2 > 2e-6:                                          # where mode 6 does run the two-piece product: the default is NOT that product
        assert _rel(dw, dy.double().t() @ x.double()) < 0.3 * e2


@pytest.mark.parametrize('B,H,W,C', [(4, 64, 35, 64), (3, 32, 18, 128), (5, 16, 9, 256), (2, 7, 5, 64), (1, 4, 4, 128), (128, 32, 18, 128), (16, 64, 35, 64)])
def test_wgrad_three_piece_planes_are_fp32_class(B, H, W, C):
    """3x3 / stride-1 / pad-1 weight gradient on the plane kernel with three pieces (64 x 32 blocks, twelve waves = 2 co halves x 3 tap rows x
    2 k halves, two partial slabs per workgroup): every trunk width, tile-aligned and ragged pixel counts (48-pixel tiles with 2 + 1 chunks per
    k half at 16 x 9), accumulate-into-.grad form -- against float64 autograd at the accuracy of the exact fp32 kernel."""
    torch.manual_seed(3)
    x = torch.randn(B, H, W, C, device=DEV)
    dy = torch.randn(B, H, W, C, device=DEV)
    w = torch.zeros(C, 3, 3, C, device=DEV)
    assert we.wgrad_planes_ok(x, w, 1, 1)
    xp, dp = ops.to_planes(x), ops.to_planes(dy)
    assert xp.shape[0] == 3
    got = we.conv_wgrad_planes(xp, dp, w, x.shape)             # logical OIHW
    xx = x.double().permute(0, 3, 1, 2)
    ww = torch.zeros(C, C, 3, 3, dtype=torch.float64, device=DEV, requires_grad=True)
    y = torch.nn.functional.conv2d(xx, ww, padding=1)
    (gref,) = torch.autograd.grad(y, ww, dy.double().permute(0, 3, 1, 2))
    lib.ha2g_gemm_set_mode(0)
    exact = we.conv_wgrad(x, dy, w, 1, 1)
    lib.ha2g_gemm_set_mode(70)
    e3, e32 = _rel(got, gref), _rel(exact, gref)
    assert e3 < 3e-6 and e3 < 1.5 * e32 + 3e-7, (e3, e32)
    base = torch.randn(C, C, 3, 3, device=DEV).contiguous(memory_format=torch.channels_last)
    tgt = base.clone(memory_format=torch.channels_last)
    assert we.conv_wgrad_planes(xp, dp, w, x.shape, into=tgt) is None
    assert _rel(tgt.double() - base.double(), gref) < 5e-6


@pytest.mark.parametrize('tile', [0, 1, 3, 4, 5, 6])
@pytest.mark.parametrize('B,H,W,Cin,Cout,k,stride,relu', [(4, 64, 35, 64, 64, 3, 1, True), (3, 32, 18, 128, 128, 3, 1, False), (5, 16, 9, 256, 256, 3, 1, True),
                                                          (4, 128, 70, 32, 64, 3, 2, True), (3, 64, 35, 64, 128, 3, 2, True), (2, 32, 18, 128, 256, 3, 2, False),
                                                          (4, 128, 70, 32, 64, 1, 2, False), (3, 64, 35, 64, 128, 1, 2, False), (1, 7, 5, 64, 64, 3, 1, False),
                                                          (128, 32, 18, 128, 128, 3, 1, True)])
def test_forward_convolution_on_three_piece_planes_is_fp32_class(B, H, W, Cin, Cout, k, stride, relu, tile):
    """The forward convolutions of trunk layers 2-4 (nn.Conv2d of SEBasicBlock, ResNetBlocks.py:24-29; the stride-2 3x3 and 1x1 downsample of a
    layer's first block) on producer-written three-piece planes: against float64 at the accuracy of the fp32 MFMA implicit GEMM."""
    torch.manual_seed(8)
    pad = 1 if k == 3 else 0
    x = torch.randn(B, H, W, Cin, device=DEV)
    w = torch.randn(Cout, k, k, Cin, device=DEV) * 0.05
    assert we.fwd_planes_ok(w, stride, pad)
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), stride=stride, padding=pad).permute(0, 2, 3, 1)
    if relu:
        ref = torch.relu(ref)
    lib.ha2g_conv_planes_tile3(tile)
    try:
        got = we.conv_fwd_planes(ops.to_planes(x, 3), ops.to_planes(w, 3), x.shape, stride, pad, ops.ACT_RELU if relu else ops.ACT_NONE)
    finally:
        lib.ha2g_conv_planes_tile3(0)
    exact = we.conv_fwd(x, w, None, stride, pad, ops.ACT_RELU if relu else ops.ACT_NONE)
    e3, e32 = _rel(got, ref), _rel(exact, ref)
    assert got.shape == exact.shape and e3 < 3e-6 and e3 < 1.5 * e32 + 2e-7, (e3, e32)


@pytest.mark.parametrize('M,N,K,ta,tb', [(13056, 900, 600, False, True), (4352, 900, 108, False, True), (4352, 600, 900, False, False),
                                         (900, 600, 4352, True, False), (900, 108, 13056, True, False), (4352, 150, 300, False, True),
                                         (1000, 300, 77, False, True), (4352, 300, 600, False, False), (300, 600, 4352, True, False)])
def test_dense_products_on_three_piece_planes(M, N, K, ta, tb):
    """ops.gemm routes large products to the plane GEMM (ha2g_gemm_planes_np_f32: K-contiguous zero-padded piece planes through
    ha2g_f32_to_planes_2d_np -- transposing where k is an operand's slow dimension --, the quantisation-free plane kernel, split-K slabs for
    small tile grids): every operand layout of the step (forward y = x W^T, dX = dY W, dW = dY^T X with its bias-gradient column sums),
    K not a multiple of 32, bias / activation / accumulate epilogues -- against float64 at the accuracy of the fp32 MFMA GEMM."""
    torch.manual_seed(9)
    a = torch.randn((K, M) if ta else (M, K), device=DEV)
    b = torch.randn((N, K) if tb else (K, N), device=DEV) * 0.05
    ref = (a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double())
    old = ops.PLANE_GEMM_MIN_FLOP
    ops.PLANE_GEMM_MIN_FLOP = 0.0
    try:
        aligned = all(t.stride(0) % 4 == 0 for t in (a, b))
        assert ops._plane_gemm_ok(a, b, M, N, K, ta, tb, 1.0, ops.ACT_NONE, None) == (min(M, N) >= 128 and K >= 64 and aligned)
        got = ops.gemm(a, b, transa=ta, transb=tb)
        ops.PLANE_GEMM = False
        exact_mode = ops.gemm(a, b, transa=ta, transb=tb)
        ops.PLANE_GEMM = True
        e3, e32 = _rel(got, ref), _rel(exact_mode, ref)
        assert e3 < 2e-6 and e3 < 3.0 * e32 + 3e-7, (e3, e32)
        if not ta:
            bias = torch.randn(N, device=DEV)
            base = torch.randn(M, N, device=DEV)
            got2 = ops.gemm(a, b, transa=ta, transb=tb, out=base.clone(), beta=1.0, bias=bias, act=ops.ACT_LEAKY)
            v = ref + bias.double() + base.double()        # act(a b + bias + beta out): the convention of ha2g_gemm_f32
            ref2 = torch.where(v > 0, v, 0.01 * v)
            assert _rel(got2, ref2) < 3e-6
        else:
            cs = torch.randn(M, device=DEV)
            cs0 = cs.clone()
            tgt = torch.randn(M, N, device=DEV)
            t0 = tgt.clone()
            ops.gemm(a, b, transa=True, out=tgt, beta=1.0, colsum_out=cs, colsum_beta=1.0)
            assert _rel(tgt.double() - t0.double(), ref) < 3e-6
            assert _rel(cs.double() - cs0.double(), a.double().sum(0)) < 3e-6
    finally:
        ops.PLANE_GEMM_MIN_FLOP = old
        ops.PLANE_GEMM = True


@pytest.mark.parametrize('shape', [(2, 128, 70), (3, 9, 37)])
def test_c32_three_piece_kernel_forms_are_fp32_class(shape):
    """The three forms of the 32-channel three-piece direct convolution (conv_c32.hip): anti-phase (default, 16x16x32 tiles), prefetching and first form
    (32x32x16 tiles) -- forward with ReLU and accumulating data gradient against float64; the two 32x32x16 forms run the same MFMA chain: bit-identical."""
    import torch.nn.functional as F
    N, H, W = shape
    g = torch.Generator().manual_seed(11 + shape[2])
    x = torch.randn(N, 32, H, W, generator=g)
    w = torch.randn(32, 32, 3, 3, generator=g) * 0.08
    dy = torch.randn(N, 32, H, W, generator=g)
    base = torch.randn(N, 32, H, W, generator=g)
    xg, dyg, wg = (t.permute(0, 2, 3, 1).contiguous().to(DEV) for t in (x, dy, w))
    ref_y = F.relu(F.conv2d(x.double(), w.double(), padding=1))
    ref_dx = base.double() + F.conv_transpose2d(dy.double(), w.double(), padding=1)
    outs = {}
    try:
        for form, sw in (('anti-phase', 1), ('prefetch', 33), ('first', 32)):
            lib.ha2g_conv_c32_prefetch(sw)
            y = we.conv_fwd(xg, wg, None, 1, 1, we.ACT_RELU)
            acc = base.permute(0, 2, 3, 1).contiguous().to(DEV)
            dx = we.conv_dgrad(dyg, wg, (N, H, W, 32), 1, 1, out=acc, beta=1.0)
            outs[form] = (y.permute(0, 3, 1, 2).double().cpu(), dx.permute(0, 3, 1, 2).double().cpu())
    finally:
        lib.ha2g_conv_c32_prefetch(1)
    for form, (y, dx) in outs.items():
        assert float((y - ref_y).abs().max() / ref_y.abs().max()) < 3e-6, form
        assert float((dx - ref_dx).abs().max() / ref_dx.abs().max()) < 3e-6, form
    assert torch.equal(outs['prefetch'][0], outs['first'][0]) and torch.equal(outs['prefetch'][1], outs['first'][1])


@pytest.mark.parametrize('B,H,W,C,Cout', [(4, 64, 35, 64, 64), (3, 64, 35, 64, 64), (4, 32, 18, 128, 128), (5, 16, 9, 256, 256), (2, 13, 11, 64, 128),
                                          (1, 7, 5, 64, 64), (7, 16, 9, 256, 256), (2, 64, 63, 64, 64), (128, 32, 18, 128, 128)])
def test_patch_resident_kernel_is_bit_identical_to_the_q_kernel(B, H, W, C, Cout):
    """pconv_r_kernel (round 5: the tile's patch staged once per 32-channel slice, the nine taps read it at shifted addresses) against pconv_q_kernel
    (every tap's A tile re-staged): same k order, same six products smallest first, same accumulators => torch.equal -- forward (with and without the
    fused ReLU) and data gradient (beta = 0 and accumulate), the three trunk geometries, odd image counts (an idle second group), images that are
    not a whole number of tiles, a 63-wide layer-2 image (the T = 62 spectrogram) -- and against float64 at the three-piece bound."""
    from ha2g_amd import wav_engine as we
    torch.manual_seed(11)
    x = torch.randn(B, H, W, C, device=DEV)
    w = torch.randn(Cout, 3, 3, C, device=DEV) * 0.05                          # OHWI
    xp, wp = ops.to_planes(x, 3), ops.to_planes(w.contiguous(), 3)
    dy = torch.randn(B, H, W, Cout, device=DEV)
    dyp = ops.to_planes(dy, 3)
    base = torch.randn(B, H, W, C, device=DEV)

    def run():
        y0 = we.conv_fwd_planes(xp, wp, x.shape, 1, 1, we.ACT_NONE)
        y1 = we.conv_fwd_planes(xp, wp, x.shape, 1, 1, we.ACT_RELU)
        d0 = we.conv_dgrad_planes(dyp, w, (B, H, W, C), 1, 1)
        d1 = we.conv_dgrad_planes(dyp, w, (B, H, W, C), 1, 1, out=base.clone(), beta=1.0)
        torch.cuda.synchronize()
        return y0, y1, d0, d1
    try:
        lib.ha2g_conv_planes_tile3(8)                                          # the q kernel
        ref = run()
    finally:
        lib.ha2g_conv_planes_tile3(0)
    got = run()                                                                # default: the patch-resident kernel where it serves the geometry
    for a, b_, name in zip(got, ref, ('fwd', 'fwd+relu', 'dgrad', 'dgrad beta=1')):
        assert torch.equal(a, b_), (name, float((a - b_).abs().max()))
    y64 = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    assert float((got[0].double() - y64).abs().max() / y64.abs().max()) < 3e-6
    d64 = torch.nn.functional.conv_transpose2d(dy.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    assert float((got[2].double() - d64).abs().max() / d64.abs().max()) < 3e-6


@pytest.mark.parametrize('relu', [False, True])
@pytest.mark.parametrize('B,H,W,C,Cout,k,stride', [(4, 64, 35, 64, 64, 3, 1), (3, 64, 35, 64, 64, 3, 1), (4, 32, 18, 128, 128, 3, 1), (5, 16, 9, 256, 256, 3, 1),
                                                   (2, 13, 11, 64, 128, 3, 1), (1, 7, 5, 64, 64, 3, 1), (2, 64, 63, 64, 64, 3, 1), (128, 32, 18, 128, 128, 3, 1),
                                                   (4, 64, 35, 32, 64, 3, 2), (4, 64, 35, 32, 64, 1, 2), (3, 32, 18, 64, 128, 3, 2), (5, 16, 9, 128, 256, 1, 2),
                                                   (128, 64, 35, 32, 64, 1, 2)])
def test_batchnorm_statistics_from_the_convolution_epilogue(B, H, W, C, Cout, k, stride, relu):
    """ha2g_conv2d_fwd_planes_np_stats_f32: the forward plane convolution leaves per-tile column sums of its stored output behind and
    ha2g_bn_stats_finalize_f32 turns them into the BatchNorm's mean / invstd / running statistics (conv -> [ReLU] -> BN, ResNetBlocks.py:24-29):
    (a) the output is the bits of the plain launch, (b) mean / var agree with float64 statistics of that output to 1e-6 of the channel scale -- the
    accuracy of the separate column pass (ha2g_bn_stats_f32), which is run beside it -- (c) the running statistics receive the same update.
    3x3 / stride 1: the patch-resident kernel's epilogue; stride 2, 1x1 and the 63-wide image: the q kernel's."""
    from ha2g_amd import wav_engine as we
    torch.manual_seed(5)
    pad = 1 if k == 3 else 0
    x = torch.randn(B, H, W, C, device=DEV)
    w = torch.randn(Cout, k, k, C, device=DEV) * 0.05
    xp, wp = ops.to_planes(x, 3), ops.to_planes(w.contiguous(), 3)
    act = we.ACT_RELU if relu else we.ACT_NONE
    y0 = we.conv_fwd_planes(xp, wp, x.shape, stride, pad, act)
    y1, st = we.conv_fwd_planes(xp, wp, x.shape, stride, pad, act, stats=True)
    nblk = lib.ha2g_conv2d_fwd_planes_stat_blocks(B, H, W, C, Cout, k, k, stride, pad)
    assert st is not None and st[1] == nblk > 0
    assert torch.equal(y0, y1)
    rows = y0.numel() // Cout
    rm0, rv0 = torch.randn(Cout, device=DEV), torch.rand(Cout, device=DEV) + 0.5
    rm_a, rv_a, rm_b, rv_b = rm0.clone(), rv0.clone(), rm0.clone(), rv0.clone()
    mean_a, inv_a = ops.bn_stats_finalize(st[0], st[1], rows, Cout, rm_a, rv_a, 0.1, 1e-5)
    mean_b, inv_b = ops.bn_stats(y0.view(rows, Cout), rm_b, rv_b, 0.1, 1e-5)
    y64 = y0.double().view(rows, Cout)
    m64, v64 = y64.mean(0), y64.var(0, unbiased=False)
    scale = float(y64.abs().max())
    for got in (mean_a, mean_b):
        assert float((got.double() - m64).abs().max()) < 1e-6 * scale, float((got.double() - m64).abs().max()) / scale
    i64 = 1.0 / torch.sqrt(v64 + 1e-5)
    for got in (inv_a, inv_b):
        assert float(((got.double() - i64) / i64).abs().max()) < 2e-6, float(((got.double() - i64) / i64).abs().max())
    assert float((rm_a - rm_b).abs().max()) < 1e-6 * scale and float(((rv_a - rv_b) / rv_b).abs().max()) < 2e-6
    # the partial sums themselves: every tile's sums add up to the column sums
    # (a lane adds its <= 9 pixels in fp32 before the double reduction: 9 roundings of 6e-8 relative to the terms)
    s64 = st[0].sum(2)
    e1 = float(((s64[0] - y64.sum(0)).abs() / y64.abs().sum(0).clamp_min(1e-30)).max())
    e2 = float(((s64[1] - (y64 * y64).sum(0)).abs() / (y64 * y64).sum(0).clamp_min(1e-30)).max())
    assert e1 < 1e-6 and e2 < 1e-6, (e1, e2)


@pytest.mark.parametrize('geom', [(4, 32, 18, 64, 64, 3, 1), (3, 64, 35, 32, 64, 3, 2)])
def test_epilogue_statistics_of_a_channel_whose_mean_dwarfs_its_spread(geom):
    """the convolution epilogue's lane sums are SHIFTED (v - K, K = the lane's first pixel) before squaring, so a channel with |mean| >> sigma --
    here 300 sigma: a convolution whose input carries a large constant -- keeps its variance: invstd from ha2g_bn_stats_finalize_f32 equals the float64
    statistics of the stored output, and the separate column pass (shifted sums in double), at 2e-6.  (Raw fp32 sums of squares lose
    (1 + mean^2 / var) x 5e-7 of the variance in s2 / n - mean^2: ~5 % here.)  Patch-resident kernel and q kernel (stride 2)."""
    from ha2g_amd import wav_engine as we
    B, H, W, C, Cout, k, stride = geom
    torch.manual_seed(7)
    x = torch.randn(B, H, W, C, device=DEV) * 0.01 + 3.0             # sum over 9 C taps of ~3 w: a large per-channel constant, a small spread
    w = torch.randn(Cout, k, k, C, device=DEV) * 0.05 + 0.02
    xp, wp = ops.to_planes(x, 3), ops.to_planes(w.contiguous(), 3)
    y, st = we.conv_fwd_planes(xp, wp, x.shape, stride, 1, we.ACT_NONE, stats=True)
    rows = y.numel() // Cout
    # the interior only defines the regime (zero padding makes the border pixels differ): check the ratio on the stored output itself
    y64 = y.double().view(rows, Cout)
    m64, v64 = y64.mean(0), y64.var(0, unbiased=False)
    assert float((m64.abs() / v64.sqrt()).median()) > 3.0
    rm, rv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
    mean_a, inv_a = ops.bn_stats_finalize(st[0], st[1], rows, Cout, rm.clone(), rv.clone(), 0.1, 1e-5)
    mean_b, inv_b = ops.bn_stats(y.view(rows, Cout), rm.clone(), rv.clone(), 0.1, 1e-5)
    i64 = 1.0 / torch.sqrt(v64 + 1e-5)
    for got in (inv_a, inv_b):
        assert float(((got.double() - i64) / i64).abs().max()) < 2e-6, float(((got.double() - i64) / i64).abs().max())
    for got in (mean_a, mean_b):
        assert float(((got.double() - m64) / m64.abs().clamp_min(1e-3)).abs().max()) < 1e-6


def test_convolution_epilogue_statistics_are_off_where_the_kernel_cannot_write_them():
    """the eight-wave ping-pong form of the plane kernel (ha2g_conv_planes_tile3(4), an A/B form) has no statistics epilogue: the query reports 0
    blocks and the caller keeps the column pass; asking for statistics with a block count the geometry does not have is an error, not a silent skip."""
    try:
        lib.ha2g_conv_planes_tile3(4)
        assert lib.ha2g_conv2d_fwd_planes_stat_blocks(4, 64, 35, 64, 64, 3, 3, 1, 1) == 0
    finally:
        lib.ha2g_conv_planes_tile3(0)
    x = torch.randn(2, 16, 9, 64, device=DEV)
    w = torch.randn(64, 3, 3, 64, device=DEV)
    xp, wp = ops.to_planes(x, 3), ops.to_planes(w, 3)
    y = torch.empty(2, 16, 9, 64, device=DEV)
    part = torch.empty(2, 64, 7, dtype=torch.float64, device=DEV)
    rc = lib.ha2g_conv2d_fwd_planes_np_stats_f32(xp.data_ptr(), xp.stride(0), wp.data_ptr(), wp.stride(0), 3, y.data_ptr(), 2, 16, 9, 64, 64, 3, 3, 1, 1, 0,
                                                 part.data_ptr(), 7, torch.cuda.current_stream().cuda_stream)
    assert rc != 0

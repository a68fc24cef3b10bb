"""Plane-based split-bf16 products (csrc/conv_planes.hip, round 3): operands pre-split into bf16 hi / lo planes by their producer, moved
global -> LDS by DMA.  Every kernel here must be BIT-IDENTICAL to the round-2 kernel it replaces (same hi / lo values, same three-MFMA product,
same k order), so the reference-derived parity fixtures cannot move: the tests compare with torch.equal.  (Replaces autograd's conv2d
backward, model/ResNetBlocks.py:24-29 under train_eval/train_hierarchy.py:264.)"""
import pytest
import torch

from ha2g_amd import ops, procedural as proc, wav_engine as we
from ha2g_amd._lib import DEFAULT_GEMM_MODE, lib

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(autouse=True)
def two_piece_mode():
    """This module pins the round-3 TWO-piece kernels (mode 6, now the labelled secondary mode) bit for bit against the kernels they replaced;
    the three-piece default of round 4 has its own accuracy tests in tests/test_gpu_np3.py."""
    lib.ha2g_gemm_set_mode(6)
    yield
    lib.ha2g_gemm_set_mode(DEFAULT_GEMM_MODE)



def test_to_planes_is_the_two_piece_bf16_split():
    torch.manual_seed(0)
    x = torch.randn(1 << 16, device=DEV) * torch.logspace(-6, 6, 1 << 16, device=DEV)
    hi, lo = ops.to_planes(x)
    assert torch.equal(hi, x.bfloat16())                                       # round-to-nearest-even
    assert torch.equal(lo, (x - hi.float()).bfloat16())
    err = (x.double() - hi.double() - lo.double()).abs() / x.double().abs()
    assert float(err.max()) <= 2.0 ** -16                                      # 16 operand mantissa bits survive


@pytest.mark.parametrize('relu_mask', [False, True])
def test_bn_bwd_planes_equal_the_split_of_the_fp32_output(relu_mask):
    torch.manual_seed(1)
    rows, C = 4 * 32 * 18, 128
    x = torch.relu(torch.randn(rows, C, device=DEV) + 0.2) if relu_mask else torch.randn(rows, C, device=DEV)
    dy = torch.randn(rows, C, device=DEV)
    mean, invstd = ops.bn_stats(x, None, None, 0.1, 1e-5)
    gamma = torch.rand(C, device=DEV) + 0.5
    dx0, dg0, db0 = ops.bn_bwd(dy, x, mean, invstd, gamma, relu_mask=relu_mask)
    dx1, dg1, db1, (hi, lo) = ops.bn_bwd(dy, x, mean, invstd, gamma, relu_mask=relu_mask, planes=True)
    assert torch.equal(dx0, dx1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
    h2, l2 = ops.to_planes(dx0)
    assert torch.equal(hi, h2) and torch.equal(lo, l2)
    _, _, _, (hi3, lo3) = ops.bn_bwd(dy, x, mean, invstd, gamma, need_dx=False, relu_mask=relu_mask, planes=True)     # planes only
    assert torch.equal(hi3, hi) and torch.equal(lo3, lo)


@pytest.mark.parametrize('B,H,W,C', [(4, 64, 35, 64), (3, 64, 35, 64), (4, 32, 18, 128), (5, 16, 9, 256), (1, 7, 5, 64), (128, 32, 18, 128)])
def test_dgrad_planes_bit_identical_to_the_split_implicit_gemm(B, H, W, C):
    """3x3 stride-1 data gradient of every trunk width the kernel serves (layers 2-4), tile-aligned and ragged pixel counts, borders smaller than
    a tile, beta = 0 and the accumulate-onto-the-residual-gradient form (beta = 1) of the identity blocks."""
    torch.manual_seed(2)
    dy = torch.randn(B, H, W, C, device=DEV)
    w = torch.randn(C, 3, 3, C, device=DEV) * 0.05                             # OHWI
    assert we.dgrad_planes_ok(w, 1, 1)
    ref = we.conv_dgrad(dy, w, (B, H, W, C), 1, 1)
    got = we.conv_dgrad_planes(ops.to_planes(dy), w, (B, H, W, C), 1, 1)
    assert torch.equal(got, ref), float((got - ref).abs().max())
    base = torch.randn(B, H, W, C, device=DEV)
    ref1 = we.conv_dgrad(dy, w, (B, H, W, C), 1, 1, out=base.clone(), beta=1.0)
    got1 = we.conv_dgrad_planes(ops.to_planes(dy), w, (B, H, W, C), 1, 1, out=base.clone(), beta=1.0)
    assert torch.equal(got1, ref1)
    # and it IS the split-bf16 product, not something looser: within the split's error of the float64 result
    x64 = torch.nn.functional.conv_transpose2d(dy.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    rel = float((got.double() - x64).abs().max() / x64.abs().max())
    assert rel < 3e-5, rel


@pytest.mark.parametrize('B,H,W,Cin,Cout,k', [(4, 128, 70, 32, 64, 3), (3, 64, 35, 64, 128, 3), (5, 32, 18, 128, 256, 3), (2, 9, 7, 64, 64, 3),
                                              (4, 128, 70, 32, 64, 1), (3, 64, 35, 64, 128, 1), (4, 32, 18, 128, 256, 1), (128, 64, 35, 64, 128, 3)])
def test_stride2_dgrad_planes_bit_identical_to_the_implicit_gemm(B, H, W, Cin, Cout, k):
    """The first block of layers 2-4: 3x3 / stride 2 / pad 1 (conv1) and 1x1 / stride 2 (downsample), even and odd input sizes, Cin = 32 (half a
    column tile).  Per parity class of output pixels only the taps that reach it are run (1 / 2 / 2 / 4 of nine): same non-zero products in the
    same order as the masked nine-tap implicit GEMM => bit-identical, at a quarter of its matrix work."""
    torch.manual_seed(5)
    pad = 1 if k == 3 else 0
    OH, OW = (H + 2 * pad - k) // 2 + 1, (W + 2 * pad - k) // 2 + 1
    dy = torch.randn(B, OH, OW, Cout, device=DEV)
    w = torch.randn(Cout, k, k, Cin, device=DEV) * 0.05
    assert we.dgrad_planes_ok(w, 2, pad)
    ref = we.conv_dgrad(dy, w, (B, H, W, Cin), 2, pad)
    got = we.conv_dgrad_planes(ops.to_planes(dy), w, (B, H, W, Cin), 2, pad)
    assert torch.equal(got, ref), float((got - ref).abs().max())
    base = torch.randn(B, H, W, Cin, device=DEV)
    ref1 = we.conv_dgrad(dy, w, (B, H, W, Cin), 2, pad, out=base.clone(), beta=1.0)
    got1 = we.conv_dgrad_planes(ops.to_planes(dy), w, (B, H, W, Cin), 2, pad, out=base.clone(), beta=1.0)
    assert torch.equal(got1, ref1)
    x64 = torch.nn.functional.conv_transpose2d(dy.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), stride=2, padding=pad,
                                               output_padding=(H - ((OH - 1) * 2 - 2 * pad + k), W - ((OW - 1) * 2 - 2 * pad + k))).permute(0, 2, 3, 1)
    rel = float((got.double() - x64).abs().max() / x64.abs().max())
    assert rel < 3e-5, rel


def test_planes_are_only_used_in_the_split_mode_and_can_be_switched_off():
    w = torch.zeros(64, 3, 3, 64, device=DEV)
    assert we.dgrad_planes_ok(w, 1, 1)
    assert we.dgrad_planes_ok(w, 2, 1) and we.dgrad_planes_ok(torch.zeros(64, 1, 1, 32, device=DEV), 2, 0)   # stride-2 blocks: parity classes
    assert not we.dgrad_planes_ok(torch.zeros(32, 3, 3, 32, device=DEV), 1, 1)  # layer1 has its direct LDS-patch kernels
    assert not we.dgrad_planes_ok(torch.zeros(32, 3, 3, 32, device=DEV), 1, 0)  # the taps' pad-0 convolutions keep the implicit GEMM
    try:
        lib.ha2g_gemm_set_mode(0)                                              # exact-fp32 mode: no split product anywhere
        assert not we.dgrad_planes_ok(w, 1, 1)
    finally:
        lib.ha2g_gemm_set_mode(6)
    try:
        lib.ha2g_conv_planes_enable(0)
        assert not we.dgrad_planes_ok(w, 1, 1)
    finally:
        lib.ha2g_conv_planes_enable(1)


@pytest.mark.parametrize('B,H,W,C', [(4, 64, 35, 64), (3, 32, 18, 128), (5, 16, 9, 256), (2, 7, 5, 64), (1, 4, 4, 128), (128, 32, 18, 128)])
def test_wgrad_planes_vs_float64_and_the_implicit_gemm(B, H, W, C):
    """3x3 / stride-1 / pad-1 weight gradient from planes: every trunk width, tile-aligned (35 and 9 tiles per image) and ragged (144 and 35
    pixels per image, an image smaller than one tile) pixel counts, accumulate-into-.grad form.  Same products as the round-2 kernel in a
    different fp32 summation order: both must sit at the split product's error (2e-5) from autograd in float64."""
    torch.manual_seed(3)
    x = torch.randn(B, H, W, C, device=DEV)
    dy = torch.randn(B, H, W, C, device=DEV)
    w = torch.zeros(C, 3, 3, C, device=DEV)
    assert we.wgrad_planes_ok(x, w, 1, 1)
    got = we.conv_wgrad_planes(ops.to_planes(x), ops.to_planes(dy), w, x.shape)             # logical OIHW
    old = we.conv_wgrad(x, dy, w, 1, 1)
    xx = x.double().permute(0, 3, 1, 2).requires_grad_(False)
    ww = torch.zeros(C, C, 3, 3, dtype=torch.float64, device=DEV, requires_grad=True)
    y = torch.nn.functional.conv2d(xx, ww, padding=1)
    (gref,) = torch.autograd.grad(y, ww, dy.double().permute(0, 3, 1, 2))
    scale = float(gref.abs().max())
    e_new, e_old = float((got.double() - gref).abs().max()) / scale, float((old.double() - gref).abs().max()) / scale
    assert e_new < 2e-5 and e_old < 2e-5, (e_new, e_old)
    # accumulate form (beta = 1 into an installed channels_last .grad)
    base = torch.randn(C, C, 3, 3, device=DEV).contiguous(memory_format=torch.channels_last)
    tgt = base.clone(memory_format=torch.channels_last)
    assert we.conv_wgrad_planes(ops.to_planes(x), ops.to_planes(dy), w, x.shape, into=tgt) is None
    assert float((tgt.double() - base.double() - gref).abs().max()) / scale < 2e-5


def test_train_step_is_bitwise_unchanged_by_the_plane_data_gradients_and_close_with_the_plane_weight_gradients():
    """One GAN-phase step of a trainer (full SE-ResNet34 audio tower, B = 4): plane-based DATA gradients leave every gradient bit-equal to the
    round-2 path; adding the plane-based WEIGHT gradients (another fp32 summation order of the same products) moves them by < 1e-5 of the norm."""
    from ha2g_amd.config import hierarchy_args
    from ha2g_testing import SpeakerVocab
    from ha2g_amd.train import HierarchyTrainer
    dev = torch.device(DEV)

    class Lang:
        n_words, word_embedding_weights = 200, None

    text, spec, target, vid = (torch.from_numpy(x).to(DEV) for x in proc.make_batch(4, 27, 200, 12, 9))

    def run(planes):
        old = we.PLANES
        we.PLANES = planes
        try:
            torch.manual_seed(4)
            ops.rng.seed(dev, 11)
            tr = HierarchyTrainer(hierarchy_args(hidden_size=32, n_layers=2), Lang(), SpeakerVocab(12), 27, dev)
            ret = tr.train_iter(11, text, spec, target, vid)
            tr.sync()
            return ret, [o.flat_g.clone() for o in tr.gen_opts + [tr.audio_opt, tr.text_opt, tr.dis_opt]]
        finally:
            we.PLANES = old
    r0, g0 = run(0)
    r1, g1 = run(9)                                  # stride-1 and stride-2 data gradients
    r3, g3 = run(11)
    assert r0 == r1 == r3
    for a, b, c in zip(g0, g1, g3):
        assert torch.equal(a, b)
        assert float((c - a).norm() / a.norm()) < 1e-5

"""SE-ResNet34 audio encoder (Hierarchical_WavEncoder) forward/backward orchestration over the HIP kernels.

Reference: scripts/model/ResNetSE34V2.py:118-218 (forward), scripts/model/ResNetBlocks.py:21-37,81-95
(SEBasicBlock / SELayer).  Activations are NHWC in HBM (channels contiguous = the K axis of the implicit
GEMM); conv weights are kept physically [Cout][KH][KW][Cin] (torch channels_last parameters with the
reference's logical OIHW shape).  The whole module is ONE autograd node: backward is written out by hand
below instead of being traced.
"""
import torch

from . import ops
from ._lib import check, lib
from .ops import ACT_NONE, ACT_RELU, ACT_SIGMOID, _p, _stream, empty, workspace

LAYERS = (3, 4, 6, 3)
FILTERS = (32, 64, 128, 256)
TAPS = (('low', 64, 2, 1), ('mid', 32, 3, 2), ('high', 16, 3, 4))      # name, channels, kernel, pixel-shuffle factor


def param_names(pose_level):
    """Flat, ordered list of parameter/buffer keys (relative to feat_extractor.) the engine consumes."""
    names = ['conv1.weight', 'conv1.bias', 'bn1']
    for li, nblk in enumerate(LAYERS):
        for j in range(nblk):
            b = 'layer%d.%d.' % (li + 1, j)
            names += [b + 'conv1.weight', b + 'bn1', b + 'conv2.weight', b + 'bn2', b + 'se.fc.0.weight', b + 'se.fc.0.bias',
                      b + 'se.fc.2.weight', b + 'se.fc.2.bias']
            if j == 0 and li > 0:
                names += [b + 'downsample.0.weight', b + 'downsample.1']
    for t, _, _, _ in TAPS:
        names += ['conv_%s.weight' % t, 'conv_%s.bias' % t, 'bn_%s' % t, 'fc_%s.weight' % t, 'fc_%s.bias' % t]
    names += ['speaker_embedding.0.weight', 'speaker_embedding.1.weight', 'speaker_embedding.1.bias', 'fc1.weight', 'fc1.bias',
              'fc2.weight', 'fc2.bias']
    return names


# ---- raw conv wrappers (NHWC tensors [N,H,W,C]; weights physical OHWI) --------------------------------------

def _ohwi(w):
    """Physical [Cout,KH,KW,Cin] view of a logical OIHW conv weight (no copy when it is channels_last)."""
    wp = w.permute(0, 2, 3, 1)
    return wp if wp.is_contiguous() else wp.contiguous()


def conv_fwd(x, w_ohwi, bias, stride, pad, act):
    N, H, W, Cin = x.shape
    Cout, KH, KW, _ = w_ohwi.shape
    OH, OW = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    y = empty(N, OH, OW, Cout, like=x)
    ops.ktimer.launch('conv2d_fwd', lambda: check(lib.ha2g_conv2d_fwd_f32(
        x.data_ptr(), w_ohwi.data_ptr(), _p(bias), y.data_ptr(), N, H, W, Cin, Cout, KH, KW, stride, pad, act, _stream())),
        2.0 * N * OH * OW * Cout * KH * KW * Cin)
    return y


FWD3 = True      # forward convolutions of trunk layers 2-4 on three-piece planes


def fwd_planes_ok(w_ohwi, stride, pad):
    """this forward convolution runs on three-piece planes (csrc/conv_planes.hip, fp32-class: six bf16 MFMAs on all 24 mantissa bits) -- only in
    the fp32-class default mode; modes 0 / 6 keep the fp32 MFMA forward of rounds 1-3"""
    Cout, KH, KW, Cin = w_ohwi.shape
    return FWD3 and lib.ha2g_gemm_bwd_pieces() == 3 and bool(lib.ha2g_conv2d_fwd_planes_supported(Cin, Cout, KH, KW, stride, pad))


RESID_EPILOGUE = True     # identity-shortcut blocks: conv1's data gradient adds the masked residual in its epilogue, dres = dout * (out > 0) is never written (round 6)
RELU_BITS = True          # the blocks' ReLU decisions (out > 0) as bits for the two-pass backward tail: 1 / 32 of the bytes of `out`, read twice per block (round 6)
IMAGE_STATS = True        # layer 1 (no statistics epilogue): bn2's statistics and the SE squeeze from ONE per-image column pass over conv2's output (round 6)
SE_FROM_STATS = True      # ... and, where the epilogue's tiles lie inside one image, the SE squeeze too: bn2's output is never materialised (block_fwd)
def _tiles_per_image(xshape, w_ohwi):
    """> 0: conv2's statistics blocks are tiles inside one image, that many per image (asked per call: host arithmetic, and the answer follows the
    library's kernel-selection switches)"""
    N, H, W, Cin = xshape
    Cout, KH, KW, _ = w_ohwi.shape
    return int(lib.ha2g_conv2d_fwd_planes_stat_tiles_per_image(N, H, W, Cin, Cout, KH, KW, 1, 1))


CONV_BN_STATS = True      # the statistics of the BatchNorm behind a forward plane convolution come out of the convolution's epilogue (no column pass)

def conv_fwd_planes(xp, wpl, xshape, stride, pad, act, stats=False):
    """conv_fwd on the piece planes of x [3, N, H, W, Cin] and of the OHWI weight [3, Cout, KH, KW, Cin]; act: ACT_NONE / ACT_RELU; fp32 output.
    stats=True: returns (y, st) with st = (partial sums [2, Cout, nblk] float64, nblk) of y for ops.bn_stats_finalize when this geometry's kernel
    carries the statistics epilogue, else st = None."""
    N, H, W, Cin = xshape
    _, Cout, KH, KW, _ = wpl.shape
    OH, OW = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    y = torch.empty(N, OH, OW, Cout, dtype=torch.float32, device=xp.device)
    if stats:
        # asked per call (host arithmetic): the answer follows the library's kernel-selection switches (ha2g_conv_planes_tile3)
        nblk = int(lib.ha2g_conv2d_fwd_planes_stat_blocks(N, H, W, Cin, Cout, KH, KW, stride, pad)) if CONV_BN_STATS else 0
        if nblk > 0:
            part = torch.empty(2, Cout, nblk, dtype=torch.float64, device=xp.device)
            ops.ktimer.launch('conv2d_fwd_planes', lambda: check(lib.ha2g_conv2d_fwd_planes_np_stats_f32(
                xp.data_ptr(), xp.stride(0), wpl.data_ptr(), wpl.stride(0), xp.shape[0], y.data_ptr(), N, H, W, Cin, Cout, KH, KW, stride, pad,
                1 if act == ACT_RELU else 0, part.data_ptr(), nblk, _stream())), 2.0 * N * OH * OW * Cout * KH * KW * Cin)
            return y, (part, nblk)
    ops.ktimer.launch('conv2d_fwd_planes', lambda: check(lib.ha2g_conv2d_fwd_planes_np_f32(
        xp.data_ptr(), xp.stride(0), wpl.data_ptr(), wpl.stride(0), xp.shape[0], y.data_ptr(), N, H, W, Cin, Cout, KH, KW, stride, pad,
        1 if act == ACT_RELU else 0, _stream())), 2.0 * N * OH * OW * Cout * KH * KW * Cin)
    return (y, None) if stats else y


def prepare_fwd_weight_planes(P):
    """three-piece planes [3, Cout, KH, KW, Cin] of the OHWI weight of every trunk convolution whose FORWARD runs on planes, in one launch
    (ha2g_f32_to_planes_multi_np) -> {parameter name: planes}"""
    import numpy as np
    items = []
    for li, nblk in enumerate(LAYERS):
        for j in range(nblk):
            b = 'layer%d.%d.' % (li + 1, j)
            first = j == 0 and li > 0
            cands = [(b + 'conv1.weight', 2 if first else 1, 1), (b + 'conv2.weight', 1, 1)] + ([(b + 'downsample.0.weight', 2, 0)] if first else [])
            for name, stride, pad in cands:
                wp = P[name].permute(0, 2, 3, 1)
                if wp.is_contiguous() and fwd_planes_ok(wp, stride, pad):
                    items.append((name, wp))
    if not items:
        return {}
    dev = items[0][1].device
    total = sum(w.numel() for _, w in items)
    buf = torch.empty(3, total, dtype=torch.bfloat16, device=dev)
    out, off = {}, 0
    xp_, pp_, ps_, ne_ = (np.empty(len(items), np.int64) for _ in range(4))
    for i, (name, w) in enumerate(items):
        n = w.numel()
        v = buf[:, off:off + n].view(3, *w.shape)
        out[name] = v
        xp_[i], pp_[i], ps_[i], ne_[i] = w.data_ptr(), v.data_ptr(), v.stride(0), n
        off += n
    for k0 in range(0, len(items), 48):
        n = min(48, len(items) - k0)
        check(lib.ha2g_f32_to_planes_multi_np(xp_[k0:].ctypes.data, pp_[k0:].ctypes.data, ps_[k0:].ctypes.data, ne_[k0:].ctypes.data, n, 3, _stream()))
    return out


def conv_dgrad(dy, w_ohwi, xshape, stride, pad, out=None, beta=0.0, resid=None):
    """resid = (dout, decision bits): out = the data gradient + (bit ? dout : 0) (ha2g_conv2d_dgrad_resid_f32; the caller checked *_resid_supported)"""
    N, H, W, Cin = xshape
    Cout, KH, KW, _ = w_ohwi.shape
    wt = empty(Cin, KH, KW, Cout, like=dy)
    check(lib.ha2g_conv2d_weight_ohwi_to_ihwo_f32(w_ohwi.data_ptr(), wt.data_ptr(), Cout, KH, KW, Cin, _stream()))
    if resid is not None:
        assert out is None
        out = empty(N, H, W, Cin, like=dy)
        check(lib.ha2g_conv2d_dgrad_resid_f32(dy.data_ptr(), wt.data_ptr(), out.data_ptr(), N, H, W, Cin, Cout, KH, KW, stride, pad, resid[0].data_ptr(),
                                              resid[1].data_ptr(), _stream()))
        return out
    if out is None:
        out = empty(N, H, W, Cin, like=dy)
    check(lib.ha2g_conv2d_dgrad_f32(dy.data_ptr(), wt.data_ptr(), out.data_ptr(), N, H, W, Cin, Cout, KH, KW, stride, pad, beta,
                                    _stream()))
    return out


def dgrad_planes_ok(w_ohwi, stride, pad):
    Cout, KH, KW, Cin = w_ohwi.shape
    return bool(PLANES & (1 if stride == 1 else 8)) and bool(lib.ha2g_conv2d_dgrad_planes_supported(Cin, Cout, KH, KW, stride, pad))


def prepare_weight_planes(P):
    """(hi, lo) bf16 planes of the transposed weight [Cin][KH][KW][Cout] of EVERY trunk convolution whose data gradient runs on the plane kernel, in
    one launch (ha2g_conv2d_weight_ihwo_planes_multi) -> {parameter name: (hi, lo)}.  Called once at the start of the tower's backward: the
    per-convolution re-layout launches (29 per step) leave its critical path."""
    import numpy as np
    items = []
    for li, nblk in enumerate(LAYERS):
        for j in range(nblk):
            b = 'layer%d.%d.' % (li + 1, j)
            first = j == 0 and li > 0
            cands = [(b + 'conv2.weight', 1, 1), (b + 'conv1.weight', 2 if first else 1, 1)] + ([(b + 'downsample.0.weight', 2, 0)] if first else [])
            for name, stride, pad in cands:
                w = _ohwi(P[name])
                if dgrad_planes_ok(w, stride, pad):
                    items.append((name, w))
    if not items:
        return {}
    dev = items[0][1].device
    sizes = [w.numel() for _, w in items]
    total = sum(sizes)
    npc = ops.pieces()
    buf = torch.empty(npc, total, dtype=torch.bfloat16, device=dev)          # every weight's piece q inside plane q of ONE buffer: equal plane stride
    out, off = {}, 0
    wp, hp, ps = (np.empty(len(items), np.int64) for _ in range(3))
    co, kk, ci = (np.empty(len(items), np.int32) for _ in range(3))
    keep = []
    for i, ((name, w), n) in enumerate(zip(items, sizes)):
        Cout, KH, KW, Cin = w.shape
        v = buf[:, off:off + n].view(npc, Cin, KH, KW, Cout)                  # strided view: piece q = v[q]
        out[name] = v
        wp[i], hp[i], ps[i] = w.data_ptr(), v.data_ptr(), v.stride(0)
        co[i], kk[i], ci[i] = Cout, KH * KW, Cin
        keep.append(w)
        off += n
    for k0 in range(0, len(items), 48):
        n = min(48, len(items) - k0)
        check(lib.ha2g_conv2d_weight_ihwo_planes_multi_np(wp[k0:].ctypes.data, hp[k0:].ctypes.data, ps[k0:].ctypes.data, co[k0:].ctypes.data,
                                                          kk[k0:].ctypes.data, ci[k0:].ctypes.data, n, npc, _stream()))
    return out


_WPLANES = [None]      # planes prepared by prepare_weight_planes for the backward in progress (name -> (hi, lo)), looked up by weight data_ptr


BN_BWD_EPILOGUE = False     # bn1's backward statistics out of the epilogue of conv2's data gradient (round 6): built, tested, NOT the default --
                            # the step is 0.3 ms SLOWER with it (36.14 vs 35.84 ms: the double-precision epilogue of the MFMA kernel costs more than the
                            # column pass it removes), DESIGN 8.1


def dgrad_bnstats_blocks(w_ohwi, xshape, stride, pad):
    """tiles per channel the data gradient's epilogue would write for the BatchNorm backward that consumes it; 0 = not served (keep the column pass)"""
    N, H, W, Cin = xshape
    Cout, KH, KW, _ = w_ohwi.shape
    if not BN_BWD_EPILOGUE:
        return 0
    return int(lib.ha2g_conv2d_dgrad_planes_stat_blocks(N, H, W, Cin, Cout, KH, KW, stride, pad))


def conv_dgrad_planes(dy_planes, w_ohwi, xshape, stride, pad, out=None, beta=0.0, bnstats=None, resid=None):
    """conv_dgrad on the pre-split bf16 planes of dy (ops.bn_bwd(..., planes=True)): the weight goes through one re-layout + split launch,
    the data gradient through the DMA-staged kernel of csrc/conv_planes.hip.  Bit-identical to conv_dgrad() in the default arithmetic mode.
    bnstats = (x_bn, mean, invstd, nblk): the output is the dy of that BatchNorm's backward and nblk = dgrad_bnstats_blocks(...) > 0 -- the epilogue
    also leaves the backward's tile sums behind; returns (out, (stat_part [2, Cin, nblk] float64, nblk)) for ops.bn_bwd(..., partials=...)."""
    N, H, W, Cin = xshape
    Cout, KH, KW, _ = w_ohwi.shape
    dyp = dy_planes                                       # [np, N, OH, OW, Cout] (or [np, rows, Cout]) bf16 piece planes
    npc = dyp.shape[0]
    wpl = _WPLANES[0].get(w_ohwi.data_ptr()) if _WPLANES[0] is not None else None
    if wpl is None or wpl.shape[0] != npc:
        wpl = weight_planes(w_ohwi, npc)
    if resid is not None:                                 # (dout, decision bits): + (bit ? dout : 0) in the epilogue (the caller checked *_resid_supported)
        assert out is None and bnstats is None and npc == 3
        out = torch.empty(N, H, W, Cin, dtype=torch.float32, device=dyp.device)
        ops.ktimer.launch('conv_dgrad_planes', lambda: check(lib.ha2g_conv2d_dgrad_planes_np_resid_f32(
            dyp.data_ptr(), dyp.stride(0), wpl.data_ptr(), wpl.stride(0), npc, out.data_ptr(), N, H, W, Cin, Cout, KH, KW, stride, pad,
            resid[0].data_ptr(), resid[1].data_ptr(), _stream())), 2.0 * N * H * W * Cin * KH * KW * Cout)
        return out
    if out is None:
        out = torch.empty(N, H, W, Cin, dtype=torch.float32, device=dyp.device)
        beta = 0.0
    if bnstats is not None:
        x_bn, mean, invstd, nblk = bnstats
        assert beta == 0.0 and npc == 3 and x_bn.is_contiguous() and tuple(x_bn.shape) == (N, H, W, Cin)
        part = torch.empty(2, Cin, nblk, dtype=torch.float64, device=dyp.device)
        ops.ktimer.launch('conv_dgrad_planes', lambda: check(lib.ha2g_conv2d_dgrad_planes_np_bnstats_f32(
            dyp.data_ptr(), dyp.stride(0), wpl.data_ptr(), wpl.stride(0), npc, out.data_ptr(), N, H, W, Cin, Cout, KH, KW, stride, pad,
            x_bn.data_ptr(), mean.data_ptr(), invstd.data_ptr(), part.data_ptr(), nblk, _stream())), 2.0 * N * H * W * Cin * KH * KW * Cout)
        return out, (part, nblk)
    if beta == 0.0 and stride == 2 and KH == 1:
        out.zero_()                                       # a 1x1 stride-2 kernel reaches one pixel in four: the kernel writes only those
    ops.ktimer.launch('conv_dgrad_planes' if stride == 1 else 'conv_dgrad_planes_s2', lambda: check(lib.ha2g_conv2d_dgrad_planes_np_f32(
        dyp.data_ptr(), dyp.stride(0), wpl.data_ptr(), wpl.stride(0), npc, out.data_ptr(), N, H, W, Cin, Cout, KH, KW, stride, pad, beta, _stream())),
        2.0 * N * H * W * Cin * KH * KW * Cout)
    return out


def weight_planes(w_ohwi, npc):
    """piece planes [np, Cin, KH, KW, Cout] of ONE transposed weight (the batched form is prepare_weight_planes)."""
    import numpy as np
    Cout, KH, KW, Cin = w_ohwi.shape
    wpl = torch.empty(npc, Cin, KH, KW, Cout, dtype=torch.bfloat16, device=w_ohwi.device)
    a = lambda v, dt: np.array([v], dt)
    wp, hp, ps = a(w_ohwi.data_ptr(), np.int64), a(wpl.data_ptr(), np.int64), a(wpl.stride(0), np.int64)
    co, kk, ci = a(Cout, np.int32), a(KH * KW, np.int32), a(Cin, np.int32)
    check(lib.ha2g_conv2d_weight_ihwo_planes_multi_np(wp.ctypes.data, hp.ctypes.data, ps.ctypes.data, co.ctypes.data, kk.ctypes.data, ci.ctypes.data,
                                                      1, npc, _stream()))
    return wpl


def conv_wgrad(x, dy, w_ohwi, stride, pad, into=None):
    """-> gradient as a logical OIHW tensor with channels_last (OHWI) memory, matching the parameter.
    into = the parameter's .grad (logical OIHW, physical OHWI): accumulate there (beta = 1) and return None."""
    N, H, W, Cin = x.shape
    Cout, KH, KW, _ = w_ohwi.shape
    beta = 0.0
    if into is not None:
        dwp = into.permute(0, 2, 3, 1)
        if dwp.is_contiguous():
            dw, beta = dwp, 1.0
        else:
            into = None
    if into is None:
        dw = empty(Cout, KH, KW, Cin, like=x)
    ws = workspace(x.device)
    need = lib.ha2g_conv2d_wgrad_workspace_bytes(N, H, W, Cin, Cout, KH, KW, stride, pad)
    assert need <= ws.numel() * 4, 'wgrad workspace %d > %d' % (need, ws.numel() * 4)
    check(lib.ha2g_conv2d_wgrad_f32(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), N, H, W, Cin, Cout, KH, KW, stride, pad, beta,
                                    ws.data_ptr(), ws.numel() * 4, _stream()))
    return None if into is not None else dw.permute(0, 3, 1, 2)


def wgrad_planes_ok(x, w_ohwi, stride, pad, hw=None):
    """x: the convolution's NHWC input (or hw = (H, W) of it when the tensor does not exist yet)"""
    Cout, KH, KW, Cin = w_ohwi.shape
    H, W = hw if hw is not None else (x.shape[1], x.shape[2])
    return bool(PLANES & 2) and bool(lib.ha2g_conv2d_wgrad_planes_supported(H, W, Cin, Cout, KH, KW, stride, pad))


def conv_wgrad_planes(x_planes, dy_planes, w_ohwi, xshape, into=None):
    """conv_wgrad (3x3 / stride 1 / pad 1) on the bf16 planes of x and dy: one DMA-staged launch + the wide reduce (csrc/conv_planes.hip)."""
    N, H, W, Cin = xshape
    Cout, KH, KW, _ = w_ohwi.shape
    beta = 0.0
    if into is not None:
        dwp = into.permute(0, 2, 3, 1)
        if dwp.is_contiguous():
            dw, beta = dwp, 1.0
        else:
            into = None
    if into is None:
        dw = torch.empty(Cout, KH, KW, Cin, dtype=torch.float32, device=x_planes[0].device)
    ws = workspace(x_planes[0].device)
    assert lib.ha2g_conv2d_wgrad_planes_workspace_bytes(N, H, W, Cin, Cout) <= ws.numel() * 4
    assert x_planes.shape[0] == dy_planes.shape[0], 'conv_wgrad_planes: x and dy must carry the same number of pieces'
    ops.ktimer.launch('conv_wgrad_planes', lambda: check(lib.ha2g_conv2d_wgrad_planes_np_f32(
        x_planes.data_ptr(), x_planes.stride(0), dy_planes.data_ptr(), dy_planes.stride(0), x_planes.shape[0], dw.data_ptr(), N, H, W, Cin, Cout, KH, KW,
        1, 1, beta, ws.data_ptr(), ws.numel() * 4, _stream())), 2.0 * N * H * W * Cin * KH * KW * Cout)
    return None if into is not None else dw.permute(0, 3, 1, 2)


def _rows(t):
    return t.view(-1, t.shape[-1])


class _BN:
    """Parameters of one BatchNorm2d: (gamma, beta, running_mean, running_var, num_batches_tracked)."""
    __slots__ = ('gamma', 'beta', 'rm', 'rv', 'nbt')

    def __init__(self, gamma, beta, rm, rv, nbt):
        self.gamma, self.beta, self.rm, self.rv, self.nbt = gamma, beta, rm, rv, nbt


_TRAINING = [True]
_NBT_PENDING = []
_FWD_PLANES = [False]      # this forward will be back-propagated through the plane-based weight gradients: producers also write bf16 planes
# this forward will be back-propagated at all.  torch.is_grad_enabled() cannot tell: inside autograd.Function.forward it is always False.
# WavEncoderFunction.forward sets it from `training and any(ctx.needs_input_grad)`; direct block_fwd() callers (tests, tools) back-propagate.
_WILL_BWD = [True]
JOIN_DGRAD = 0             # round-6 experiment: 1 = the main stream waits for the side queue after every data gradient of the tower's backward (the
                           # BatchNorm / SE passes then run with no weight gradient beside them); 0 = one join at the end
SE_BWD_FOLD = True         # the SE backward's reduction finishes inside the excitation MLP's backward launch (round 6: -16 launches, same bits)
SE_BN2_FUSED = True    # SE backward + bn2 backward in two passes, bn2's dy never stored (ha2g_se_bn_bwd_*; A/B switch)
SE_WGRAD_BATCH = True      # ... of all blocks of the tower in ONE launch after its backward is enqueued (GradSink.flush_se; round 6)
SE_WGRAD_FUSED = True      # the SE excitation MLP's four parameter gradients in one launch (GradSink.gse)


def _bn_fwd(x, bn, pool=False, planes=False, planes_only=False, stats=None):
    """BatchNorm forward; pool=True also returns the per-image channel means of the output (the SE squeeze), fused; planes=True appends the
    (hi, lo) bf16 planes of the output (written by the same apply pass); stats = the partial sums x's producer left behind (conv_fwd_planes)."""
    x2 = _rows(x)
    if not _TRAINING[0]:                                    # module.eval(): running statistics, no update
        mean, invstd = bn.rm, ops.eltwise(ops.OP_RSQRT_EPS, bn.rv, alpha=1e-5)
    else:
        if stats is not None:
            mean, invstd = ops.bn_stats_finalize(stats[0], stats[1], x2.shape[0], x2.shape[1], bn.rm, bn.rv, 0.1, 1e-5)
        else:
            mean, invstd = ops.bn_stats(x2, bn.rm, bn.rv, 0.1, 1e-5)
        if bn.nbt is not None:
            _NBT_PENDING.append(bn.nbt)                    # num_batches_tracked += 1, batched into one launch per forward
    if pool:
        y, pooled = ops.bn_apply_pool(x, mean, invstd, bn.gamma, bn.beta)
        return y, mean, invstd, pooled
    if planes:
        npc = 3 if planes == 3 else 2
        y = torch.empty_like(x2) if not planes_only else None
        pl = torch.empty((npc,) + tuple(x.shape), dtype=torch.bfloat16, device=x.device)     # the x operand of the next convolution (forward / weight gradient)
        check(lib.ha2g_bn_apply_planes_np_f32(x2.data_ptr(), mean.data_ptr(), invstd.data_ptr(), bn.gamma.data_ptr(), bn.beta.data_ptr(), _p(y),
                                              pl.data_ptr(), pl.stride(0), npc, x2.shape[0], x2.shape[1], ACT_NONE, _stream()))
        return (y.view(x.shape) if y is not None else None), mean, invstd, pl
    y = ops.bn_apply(x2, mean, invstd, bn.gamma, bn.beta).view(x.shape)
    return y, mean, invstd


def _pixel_shuffle(x, r, inverse=False, shape=None):
    if not inverse:
        N, H, W, C = x.shape
        out = empty(N, H * r, W * r, C // (r * r), like=x)
        check(lib.ha2g_pixel_shuffle_f32(x.data_ptr(), out.data_ptr(), N, H, W, C // (r * r), r, 0, _stream()))
        return out
    N, H, W, C = shape                                   # unshuffled shape
    out = empty(N, H, W, C, like=x)
    check(lib.ha2g_pixel_shuffle_f32(x.data_ptr(), out.data_ptr(), N, H, W, C // (r * r), r, 1, _stream()))
    return out


def _tap_pack(x, inverse=False, shape=None):
    if not inverse:
        N, H, W, C = x.shape
        out = empty(N * W, C * H, like=x)
        check(lib.ha2g_nhwc_to_nwch_f32(x.data_ptr(), out.data_ptr(), N, H, W, C, 0, _stream()))
        return out
    N, H, W, C = shape
    out = empty(N, H, W, C, like=x)
    check(lib.ha2g_nhwc_to_nwch_f32(x.data_ptr(), out.data_ptr(), N, H, W, C, 1, _stream()))
    return out


import os as _os
# producer-side bf16 hi / lo planes for the backward convolutions (round 3).  bit 0: data gradients (bit-identical to the round-2 kernels),
# bit 1: weight gradients (same products, different fp32 summation order); bit 2: the x operand of those weight gradients is written as planes by
# the FORWARD producers (bn1's apply pass, the block's output pass) instead of being split by a streaming pass on the backward's side stream;
# bit 3: the stride-2 data gradients (conv1 / downsample of a layer's first block) by parity class; HA2G_PLANES=0 is the round-2 path
PLANES = int(_os.environ.get('HA2G_PLANES', '11'))
SIDE_WGRAD = True            # the tower's convolution weight gradients on ops.side's stream (False: in line; same kernels, bit-identical -- tested)
# bf16-storage mode of the trunk (BASELINE config 5; wav_b16.py): opt-in -- set_b16(True) or bench.py --bf16
B16 = [False]


def set_b16(on):
    """Store the trunk's activations and activation gradients as bf16 (wav_b16.py); returns the previous setting."""
    prev = B16[0]
    B16[0] = bool(on)
    return prev

SIDE_FC_WGRAD = True
WGRAD_AFTER = False          # fork a convolution's weight gradient AFTER its data gradient was enqueued (round 3 A/B: the contention only moves)
WPLANES_MULTI = True      # all weight planes of the tower's backward in one launch


SE_MLP_FUSED = True


def se_mlp_bwd(dsc, h1, w2, w0, HW):
    """Data path of the SE excitation MLP's backward: (dh1 [N,R], dpool [N,C]) from dsc [N,C] (gradient of the gate's pre-activation), the hidden
    activations h1 [N,R], fc.2.weight [C,R] and fc.0.weight [R,C] -- one launch (ha2g_se_mlp_bwd_f32) instead of GEMM + ReLU' + GEMM."""
    N, C = dsc.shape
    R = h1.shape[1]
    if SE_MLP_FUSED and w2.is_contiguous() and w0.is_contiguous() and lib.ha2g_se_mlp_bwd_supported(C, R):
        dh1, dpool = torch.empty_like(h1), torch.empty_like(dsc)
        check(lib.ha2g_se_mlp_bwd_f32(dsc.data_ptr(), h1.data_ptr(), w2.data_ptr(), w0.data_ptr(), dh1.data_ptr(), dpool.data_ptr(), N, C, R,
                                      1.0 / HW, _stream()))
        return dh1, dpool
    dh1 = ops.eltwise(ops.OP_RELU_BWD, ops.gemm(dsc, w2), h1)
    return dh1, ops.gemm(dh1, w0, alpha=1.0 / HW)


class GradSink:
    """Where the backward kernels put parameter gradients: straight into a parameter's installed `.grad` buffer when
    there is one (beta = 1 epilogues), else into the dict G (name -> grad; BatchNorm: (dgamma, dbeta)).  The convolution weight
    gradients are enqueued on ops.side's stream: call join(device) before reading any gradient."""

    def __init__(self, P):
        self.P, self.G = P, {}

    @staticmethod
    def tgt(t):                                           # a parameter's installed .grad buffer, if kernels may add into it
        g = ops._grad_target(t)
        return g if (g is not None and (g.is_contiguous() or g.dim() == 4)) else None

    def gw(self, name, a, b_):                            # dW (+)= a^T b_
        t = self.tgt(self.P[name])
        if t is not None and t.is_contiguous():
            ops.gemm(a, b_, transa=True, out=t, beta=1.0)
        else:
            self.G[name] = ops.gemm(a, b_, transa=True)

    def gwb(self, wname, bname, a, b_):                   # dW (+)= a^T b_ and db (+)= column sums of a, one launch (ha2g_gemm_wgrad_bias_f32)
        if SIDE_FC_WGRAD and SIDE_WGRAD and ops.side.enabled and a.is_cuda:
            with ops.side.section(a.device):              # off the data-gradient chain, like the convolution weight gradients
                st = ops.cur_stream(a.device)
                a.record_stream(st); b_.record_stream(st)
                self._gwb(wname, bname, a, b_)
            self.forked = True
        else:
            self._gwb(wname, bname, a, b_)

    def gse(self, b, dsc, h1, dh1, pooled):
        """the four parameter gradients of a block's SE excitation MLP: ONE launch (ha2g_se_mlp_wgrad_f32) on the side stream when all four accumulate into
        installed .grad buffers, else the two generic weight-gradient GEMMs"""
        names = [b + 'se.fc.2.weight', b + 'se.fc.2.bias', b + 'se.fc.0.weight', b + 'se.fc.0.bias']
        tg = [self.tgt(self.P[n]) for n in names]
        N, C = dsc.shape
        R = h1.shape[1]
        if not (SE_WGRAD_FUSED and all(t is not None and t.is_contiguous() for t in tg) and dsc.is_cuda and lib.ha2g_se_mlp_bwd_supported(C, R)):
            self.gwb(names[0], names[1], dsc, h1)
            self.gwb(names[2], names[3], dh1, pooled)
            return
        side_on = SIDE_FC_WGRAD and SIDE_WGRAD and ops.side.enabled
        if SE_WGRAD_BATCH:
            # kept until the trunk's backward is enqueued: one launch for all blocks (flush_se, from join())
            if not hasattr(self, 'se_jobs'):
                self.se_jobs = []
            if self.se_jobs and (self.se_jobs[0][0].shape[0] != N or len(self.se_jobs) == 16):
                self.flush_se()
            self.se_jobs.append((dsc, h1, dh1, pooled, tg))
            return
        with (ops.side.section(dsc.device) if side_on else ops._null()):
            if side_on:
                st = ops.cur_stream(dsc.device)
                for t in (dsc, h1, dh1, pooled):
                    t.record_stream(st)
            check(lib.ha2g_se_mlp_wgrad_f32(dsc.data_ptr(), h1.data_ptr(), dh1.data_ptr(), pooled.data_ptr(), tg[0].data_ptr(), tg[1].data_ptr(),
                                            tg[2].data_ptr(), tg[3].data_ptr(), N, C, R, _stream()))
        if side_on:
            self.forked = True

    def _gwb(self, wname, bname, a, b_):
        tw, tb = self.tgt(self.P[wname]), self.tgt(self.P[bname])
        if ops.FUSE_BIAS_GRAD and tw is not None and tw.is_contiguous() and tb is not None:
            ops.gemm(a, b_, transa=True, out=tw, beta=1.0, colsum_out=tb, colsum_beta=1.0)
        elif ops.FUSE_BIAS_GRAD and tw is None and tb is None:
            db = torch.empty(a.shape[1], dtype=torch.float32, device=a.device)
            self.G[wname] = ops.gemm(a, b_, transa=True, colsum_out=db)
            self.G[bname] = db
        else:
            self.gw(wname, a, b_)
            self.gb(bname, a)

    def gb(self, name, a):                                # db (+)= column sums of a
        t = self.tgt(self.P[name])
        if t is not None:
            ops.colsum(a, out=t, beta=1.0)
        else:
            self.G[name] = ops.colsum(a)

    def gconv(self, name, xin, dyc, w_ohwi, stride, pad, dy_planes=None, x_planes=None):
        """dy_planes: the (hi, lo) planes of dyc when its producer wrote them; x_planes: those of xin when the forward wrote them (else x is
        split here, once, by a streaming pass on the side stream) -- the plane-based weight gradient reads both by DMA."""
        if dy_planes is not None and wgrad_planes_ok(xin, w_ohwi, stride, pad):
            return self._gconv_planes(name, xin, dy_planes, w_ohwi, x_planes)
        return self._gconv(name, xin, dyc, w_ohwi, stride, pad)

    def _gconv_planes(self, name, xin, dy_planes, w_ohwi, x_planes=None):
        side_on = SIDE_WGRAD and ops.side.enabled and xin.is_cuda
        ctx = ops.side.section(xin.device) if side_on else ops._null()
        with ctx:
            if side_on:
                st = ops.cur_stream(xin.device)
                xin.record_stream(st); dy_planes.record_stream(st)
            if x_planes is not None and side_on:
                x_planes.record_stream(st)
            xp = x_planes if (x_planes is not None and x_planes.shape[0] == dy_planes.shape[0]) else ops.to_planes(xin, dy_planes.shape[0])
            r = conv_wgrad_planes(xp, dy_planes, w_ohwi, xin.shape, into=self.tgt(self.P[name]))
            if r is not None and side_on:
                r.record_stream(torch.cuda.default_stream(xin.device))
        if side_on:
            self.forked = True
        if r is not None:
            self.G[name] = r

    def _gconv(self, name, xin, dyc, w_ohwi, stride, pad):
        assert xin.dtype == torch.float32, 'GradSink._gconv: the fp32 activation was not materialised (planes-only producer) but a non-plane weight gradient needs it'
        # the convolution weight gradients only feed the optimizer: on the side stream they overlap the data-gradient chain -- MFMA work
        # beside the bandwidth-bound BatchNorm / SE passes of the main stream.  The operands are handed to the side stream's allocator
        # bookkeeping (record_stream) because the caller drops them before the join at the end of the tower's backward.
        if SIDE_WGRAD and ops.side.enabled and xin.is_cuda:
            with ops.side.section(xin.device):
                st = ops.cur_stream(xin.device)
                xin.record_stream(st); dyc.record_stream(st)
                r = conv_wgrad(xin, dyc, w_ohwi, stride, pad, into=self.tgt(self.P[name]))
                if r is not None:
                    r.record_stream(torch.cuda.default_stream(xin.device))
            self.forked = True
        else:
            r = conv_wgrad(xin, dyc, w_ohwi, stride, pad, into=self.tgt(self.P[name]))
        if r is not None:
            self.G[name] = r

    def flush_se(self):
        """the SE excitation MLPs' parameter gradients gse() collected: ONE launch (ha2g_se_mlp_wgrad_multi_f32), on the side stream when that is on"""
        jobs = getattr(self, 'se_jobs', None)
        if not jobs:
            return
        import numpy as np
        self.se_jobs = []
        dev = jobs[0][0].device
        n = len(jobs)
        ptr = np.empty((8, n), np.int64)
        Cs, Rs = np.empty(n, np.int32), np.empty(n, np.int32)
        for i, (dsc, h1, dh1, pooled, tg) in enumerate(jobs):
            ptr[:, i] = (dsc.data_ptr(), h1.data_ptr(), dh1.data_ptr(), pooled.data_ptr(), tg[0].data_ptr(), tg[1].data_ptr(), tg[2].data_ptr(), tg[3].data_ptr())
            Cs[i], Rs[i] = dsc.shape[1], h1.shape[1]
        side_on = SIDE_FC_WGRAD and SIDE_WGRAD and ops.side.enabled
        with (ops.side.section(dev) if side_on else ops._null()):
            if side_on:
                st = ops.cur_stream(dev)
                for job in jobs:
                    for t in job[:4]:
                        t.record_stream(st)
            check(lib.ha2g_se_mlp_wgrad_multi_f32(n, *(ptr[k].ctypes.data for k in range(8)), Cs.ctypes.data, Rs.ctypes.data, jobs[0][0].shape[0], _stream()))
        if side_on:
            self.forked = True

    def join(self, device):
        self.flush_se()
        if getattr(self, 'forked', False):
            ops.side.join(device)
            self.forked = False

    def gbn_se(self, name, dout, out, x, sc, dpool, mean, invstd, dres, stat, planes=False, need_dx=True, mask_bits=None):
        """bn2's backward fused with the SE apply pass (ha2g_se_bn_bwd_apply_np_f32): dres <- dout * (out > 0); -> (dx fp32 NHWC or None, piece planes or
        None) of bn2's data gradient; gamma / beta gradients go where gbn() puts them.  stat = the per-image sums ha2g_se_bn_bwd_reduce_mlp_f32 left."""
        bn = self.P[name]
        N, OH, OW, C = x.shape
        tg_, tb_ = self.tgt(bn.gamma), self.tgt(bn.beta)
        acc = (tg_, tb_) if (tg_ is not None and tb_ is not None) else None
        need_dx = need_dx or not planes
        dxo = torch.empty_like(x) if need_dx else None
        pl = torch.empty(ops.pieces(), N * OH * OW, C, dtype=torch.bfloat16, device=x.device) if planes else None
        dgamma, dbeta = empty(C, like=x), empty(C, like=x)
        # algorithmic HBM bytes (bench.py roofline_se_bn): reads dout and x (+ the decision bits or `out`), writes dres (+ dx) (+ the piece planes)
        nb_ = 4.0 * x.numel() * (2 + (1 if dres is not None else 0) + (1.0 / 32 if mask_bits is not None else 1) + (1 if need_dx else 0)) + (
            2.0 * pl.shape[0] * x.numel() if planes else 0.0)
        ops.ktimer.launch('se_bn_bwd_apply', lambda: check(lib.ha2g_se_bn_bwd_apply_np_f32(
            dout.data_ptr(), out.data_ptr(), x.data_ptr(), sc.data_ptr(), dpool.data_ptr(), mean.data_ptr(), invstd.data_ptr(), bn.gamma.data_ptr(),
            _p(dres), _p(dxo), _p(pl), pl.stride(0) if planes else 0, pl.shape[0] if planes else 0, dgamma.data_ptr(), dbeta.data_ptr(),
            _p(acc[0] if acc else None), _p(acc[1] if acc else None), stat.data_ptr(), N, OH * OW, C, _p(mask_bits), _stream())), nb_)
        if acc is None:
            self.G[name] = (dgamma, dbeta)
        return dxo, pl

    def gbn(self, name, dy2, x2, mean, invstd, relu_mask=False, planes=False, need_dx=True, partials=None):
        """planes=True: -> (dx fp32 or None (need_dx=False: every consumer reads the planes), (hi, lo) bf16 planes of dx).
        partials = (stat_part, nblk): the statistics pass already happened in dy's producer (conv_dgrad_planes(..., bnstats=...))"""
        bn = self.P[name]
        tg_, tb_ = self.tgt(bn.gamma), self.tgt(bn.beta)
        acc = (tg_, tb_) if (tg_ is not None and tb_ is not None) else None
        r = ops.bn_bwd(dy2, x2, mean, invstd, bn.gamma, need_dx=need_dx or not planes, relu_mask=relu_mask, acc=acc, planes=planes, partials=partials)
        if acc is None:
            self.G[name] = (r[1], r[2])
        return (r[0], r[3]) if planes else r[0]


# ---- one SEBasicBlock (ResNetBlocks.py:21-37,81-95): conv -> ReLU -> BN -> conv -> BN -> SE -> (+ residual) -> ReLU --------

def block_fwd(x, P, b, first, xp=None, out_planes=0, wpl=None):
    """x NHWC; P: name -> tensor / _BN with keys prefixed by `b`; first = stride-2 block with the 1x1 downsample branch.
    xp = piece planes of x when its producer wrote them (3 pieces: this block's convolutions of x run on them -- forward AND weight gradient;
    2 pieces: the round-3 opt-in, weight gradient only); out_planes = pieces to write this block's output with (0 = none); wpl = forward weight
    planes (prepare_fwd_weight_planes).  Returns (out NHWC, saved tuple for block_bwd, planes of out or None)."""
    stride = 2 if first else 1
    wa, wb = _ohwi(P[b + 'conv1.weight']), _ohwi(P[b + 'conv2.weight'])
    wpl = wpl or {}
    x3 = xp is not None and xp.shape[0] == 3
    st1 = st2 = None
    if x3 and (b + 'conv1.weight') in wpl:
        c1 = conv_fwd_planes(xp, wpl[b + 'conv1.weight'], x.shape, stride, 1, ACT_RELU, stats=_TRAINING[0])
        if _TRAINING[0]:
            c1, st1 = c1
    else:
        c1 = conv_fwd(x, wa, None, stride, 1, ACT_RELU)                 # relu(conv1)
    a1p = None
    f2 = (b + 'conv2.weight') in wpl
    if f2 or (_FWD_PLANES[0] and wgrad_planes_ok(c1, wb, 1, 1)):
        # planes only: when conv2's forward, data gradient and weight gradient all read the three piece planes (which hold bn1's output exactly),
        # the fp32 tensor has no reader left -- the backward of bn1 needs its INPUT c1.  a1 is then a placeholder carrying shape and device.
        # With a backward ahead that holds only if BOTH backward convolutions of conv2 run on planes (a spectrogram longer than 72 frames makes
        # W >= 37 and the plane weight gradient unsupported; HA2G_PLANES may switch either off): otherwise the fp32 tensor is materialised too.
        bwd_on_planes = dgrad_planes_ok(wb, 1, 1) and wgrad_planes_ok(c1, wb, 1, 1)
        only = f2 and (not _WILL_BWD[0] or bwd_on_planes)
        a1, m1, s1, a1p = _bn_fwd(c1, P[b + 'bn1'], planes=3 if f2 else 2, planes_only=only, stats=st1)
        if a1 is None:
            a1 = a1p[0]                                            # bf16 piece 0: NOT the activation -- _gconv() refuses non-fp32 operands
    else:
        a1, m1, s1 = _bn_fwd(c1, P[b + 'bn1'], stats=st1)
    if f2:
        c2 = conv_fwd_planes(a1p, wpl[b + 'conv2.weight'], a1.shape, 1, 1, ACT_NONE, stats=_TRAINING[0])
        if _TRAINING[0]:
            c2, st2 = c2
    else:
        c2 = conv_fwd(a1, wb, None, 1, 1, ACT_NONE)
    bn2 = P[b + 'bn2']
    N, OH, OW, C = c2.shape
    w0, b0, w2, b2_ = (P[b + n] for n in ('se.fc.0.weight', 'se.fc.0.bias', 'se.fc.2.weight', 'se.fc.2.bias'))
    R = w0.shape[0]
    mlp1 = SE_MLP_FUSED and w0.is_contiguous() and w2.is_contiguous() and bool(lib.ha2g_se_mlp_bwd_supported(C, R))      # the excitation MLP in one launch
    h1 = sc = None
    img_parts = False
    if (st2 is None and IMAGE_STATS and SE_FROM_STATS and _TRAINING[0] and mlp1 and c2.dtype == torch.float32 and c2.is_contiguous() and c2.is_cuda
            and C % 4 == 0 and 256 % (C // 4) == 0):
        # no statistics epilogue (layer 1's 32-channel convolutions): one per-image column pass over c2 serves bn2's statistics AND the SE squeeze
        nblk = N * int(lib.ha2g_bn_image_partial_chunks(N, OH * OW))
        part = torch.empty(2 * C * nblk, dtype=torch.float64, device=c2.device)
        check(lib.ha2g_bn_image_partials_f32(c2.data_ptr(), N, OH * OW, C, part.data_ptr(), _stream()))
        st2, img_parts = (part, nblk), True
    if st2 is not None and SE_FROM_STATS and (img_parts or _tiles_per_image(a1.shape, wb) > 0):
        # conv2's epilogue left per-tile column sums of c2 behind, tiles inside one image: bn2's statistics AND the SE squeeze come from them
        # (the mean of an affine map is the affine map of the mean), the tail below applies bn2 on the fly -- b2 is never written or re-read
        m2, s2 = ops.bn_stats_finalize(st2[0], st2[1], N * OH * OW, C, bn2.rm, bn2.rv, 0.1, 1e-5)
        if bn2.nbt is not None:
            _NBT_PENDING.append(bn2.nbt)
        pooled = torch.empty(N, C, dtype=torch.float32, device=c2.device)
        if mlp1:                                               # squeeze + fc.0 + ReLU + fc.2 + sigmoid: one launch
            h1, sc = torch.empty(N, R, dtype=torch.float32, device=c2.device), torch.empty(N, C, dtype=torch.float32, device=c2.device)
            check(lib.ha2g_se_mlp_fwd_f32(None, st2[0].data_ptr(), st2[1], OH * OW, m2.data_ptr(), s2.data_ptr(), bn2.gamma.data_ptr(), bn2.beta.data_ptr(),
                                          w0.data_ptr(), b0.data_ptr(), w2.data_ptr(), b2_.data_ptr(), pooled.data_ptr(), h1.data_ptr(), sc.data_ptr(),
                                          N, C, R, _stream()))
        else:
            check(lib.ha2g_bn_pool_from_partials_f32(st2[0].data_ptr(), st2[1], N, OH * OW, C, m2.data_ptr(), s2.data_ptr(), bn2.gamma.data_ptr(),
                                                     bn2.beta.data_ptr(), pooled.data_ptr(), _stream()))
        b2 = None
    else:
        b2, m2, s2, pooled = _bn_fwd(c2, bn2, pool=True, stats=st2)      # bn2 + SE squeeze in one pass
        if mlp1:
            h1, sc = torch.empty(N, R, dtype=torch.float32, device=c2.device), torch.empty(N, C, dtype=torch.float32, device=c2.device)
            check(lib.ha2g_se_mlp_fwd_f32(pooled.data_ptr(), None, 0, 0, None, None, None, None, w0.data_ptr(), b0.data_ptr(), w2.data_ptr(), b2_.data_ptr(),
                                          None, h1.data_ptr(), sc.data_ptr(), N, C, R, _stream()))
    if h1 is None:
        h1 = ops.gemm(pooled, w0, transb=True, bias=b0, act=ACT_RELU)
        # the gate straight out of the GEMM's sigmoid epilogue; the backward takes sigma' = s (1 - s) from the stored gate
        sc = ops.gemm(h1, w2, transb=True, bias=b2_, act=ACT_SIGMOID)
    su = None
    if first:
        wd = _ohwi(P[b + 'downsample.0.weight'])
        std = None
        if x3 and (b + 'downsample.0.weight') in wpl:
            cd = conv_fwd_planes(xp, wpl[b + 'downsample.0.weight'], x.shape, 2, 0, ACT_NONE, stats=_TRAINING[0])
            if _TRAINING[0]:
                cd, std = cd
        else:
            cd = conv_fwd(x, wd, None, 2, 0, ACT_NONE)
        res, md, sd = _bn_fwd(cd, P[b + 'downsample.1'], stats=std)
    else:
        res, cd, md, sd = x, None, None, None
    out = torch.empty_like(c2)
    outp = None
    if out_planes:
        outp = torch.empty((out_planes,) + tuple(out.shape), dtype=torch.bfloat16, device=out.device)
    mb = None
    if b2 is None and RELU_BITS and SE_BN2_FUSED and _TRAINING[0] and _WILL_BWD[0] and C % 32 == 0 and c2.dtype == torch.float32:
        # the block's ReLU decisions as bits: its backward (two passes, block_bwd) reads them instead of `out`
        mb = torch.empty(c2.numel() // 32, dtype=torch.int32, device=c2.device)
        check(lib.ha2g_se_bn_scale_add_relu_mask_np_f32(c2.data_ptr(), m2.data_ptr(), s2.data_ptr(), bn2.gamma.data_ptr(), bn2.beta.data_ptr(), sc.data_ptr(),
                                                        res.data_ptr(), out.data_ptr(), _p(outp), outp.stride(0) if outp is not None else 0, out_planes or 0,
                                                        N, OH * OW, C, mb.data_ptr(), _stream()))
    elif b2 is None:
        check(lib.ha2g_se_bn_scale_add_relu_np_f32(c2.data_ptr(), m2.data_ptr(), s2.data_ptr(), bn2.gamma.data_ptr(), bn2.beta.data_ptr(), sc.data_ptr(),
                                                   res.data_ptr(), out.data_ptr(), _p(outp), outp.stride(0) if outp is not None else 0, out_planes or 0,
                                                   N, OH * OW, C, _stream()))
    elif out_planes:
        check(lib.ha2g_se_scale_add_relu_planes_np_f32(b2.data_ptr(), sc.data_ptr(), res.data_ptr(), out.data_ptr(), outp.data_ptr(), outp.stride(0),
                                                       out_planes, N, OH * OW, C, _stream()))
    else:
        check(lib.ha2g_se_scale_add_relu_f32(b2.data_ptr(), sc.data_ptr(), res.data_ptr(), out.data_ptr(), N, OH * OW, C, _stream()))
    return out, (x, c1, m1, s1, a1, c2, m2, s2, b2, pooled, h1, sc, su, cd, md, sd, out, stride, xp, a1p, mb), outp


def block_bwd(dx, saved, P, b, sink):
    """dx = d(out) NHWC -> d(x) NHWC; parameter gradients go to `sink` (GradSink)."""
    (x, c1, m1, s1, a1, c2, m2, s2, b2, pooled, h1, sc, su, cd, md, sd, out, stride, xp, a1p, mb) = saved
    N, OH, OW, C = c2.shape
    HW = OH * OW
    dout = dx.contiguous()
    ds = empty(N, C, like=c2)
    w2_, w0_ = P[b + 'se.fc.2.weight'], P[b + 'se.fc.0.weight']
    wb = _ohwi(P[b + 'conv2.weight'])
    wa = _ohwi(P[b + 'conv1.weight'])
    p2, p1 = dgrad_planes_ok(wb, 1, 1), dgrad_planes_ok(wa, stride, 1)
    f2 = not (p2 and wgrad_planes_ok(a1, wb, 1, 1))                           # someone still reads the fp32 tensor
    f1 = not (p1 and wgrad_planes_ok(x, wa, stride, 1))
    fused_tail = (SE_BN2_FUSED and SE_MLP_FUSED and c2.dtype == torch.float32 and c2.is_contiguous() and w2_.is_contiguous() and w0_.is_contiguous()
                  and lib.ha2g_se_mlp_bwd_supported(C, h1.shape[1]))
    if fused_tail:
        # SE backward + bn2 backward in two passes over (dout, out, c2): bn2's dy is never stored, its statistics come from the SE reduction's per-image sums
        bn2 = P[b + 'bn2']
        ws = ops.workspace(dout.device)
        assert lib.ha2g_se_bn_bwd_workspace_floats(N, HW, C) <= ws.numel()
        dh1, dpool = torch.empty_like(h1), torch.empty_like(ds)
        stat = torch.empty(2 * C * N, dtype=torch.float64, device=dout.device)
        ops.ktimer.launch('se_bn_bwd_reduce', lambda: check(lib.ha2g_se_bn_bwd_reduce_mlp_f32(
            dout.data_ptr(), out.data_ptr(), c2.data_ptr(), m2.data_ptr(), s2.data_ptr(), bn2.gamma.data_ptr(), bn2.beta.data_ptr(), ds.data_ptr(), N, HW, C,
            sc.data_ptr(), ws.data_ptr(), h1.data_ptr(), w2_.data_ptr(), w0_.data_ptr(), dh1.data_ptr(), dpool.data_ptr(), h1.shape[1], stat.data_ptr(), _p(mb),
            _stream())), 4.0 * c2.numel() * (2 + (1.0 / 32 if mb is not None else 1)))
        sink.gse(b, ds, h1, dh1, pooled)
        # identity shortcut + decision bits: conv1's data gradient adds the masked residual (dout where out > 0) in its epilogue -- dres is not written
        resid = None
        if RESID_EPILOGUE and mb is not None and cd is None and stride == 1 and x.shape == c2.shape:
            Nx, Hx, Wx, Cx = x.shape
            if p1:
                if lib.ha2g_conv2d_dgrad_planes_resid_supported(Nx, Hx, Wx, Cx, wa.shape[0], 3, 3, 1, 1) and ops.pieces() == 3:
                    resid = (dout, mb)
            elif lib.ha2g_conv2d_dgrad_resid_supported(Hx, Wx, Cx, wa.shape[0], wa.shape[1], wa.shape[2], 1, 1):
                resid = (dout, mb)
        dres = torch.empty_like(c2) if resid is None else None
        dc2, dc2p = sink.gbn_se(b + 'bn2', dout, out, c2, sc, dpool, m2, s2, dres, stat, planes=p2, need_dx=f2, mask_bits=mb)
    elif SE_BWD_FOLD and SE_MLP_FUSED and w2_.is_contiguous() and w0_.is_contiguous() and lib.ha2g_se_mlp_bwd_supported(C, h1.shape[1]):
        # reduction pass + (its final pass inside) the excitation MLP's backward: two launches instead of three, the same bits
        dh1, dpool = torch.empty_like(h1), torch.empty_like(ds)
        bn2 = P[b + 'bn2'] if b2 is None else None
        xsrc = c2 if b2 is None else b2
        check(lib.ha2g_se_bwd_scale_mlp_f32(dout.data_ptr(), out.data_ptr(), xsrc.data_ptr(), _p(m2 if bn2 is not None else None),
                                            _p(s2 if bn2 is not None else None), _p(bn2.gamma if bn2 is not None else None),
                                            _p(bn2.beta if bn2 is not None else None), ds.data_ptr(), N, HW, C, sc.data_ptr(),
                                            ops.workspace(dout.device).data_ptr(), h1.data_ptr(), w2_.data_ptr(), w0_.data_ptr(), dh1.data_ptr(), dpool.data_ptr(),
                                            h1.shape[1], _stream()))
        dsc = ds
    else:
        if b2 is None:                                      # the forward never wrote bn2's output: recomputed per element from c2 (the same bits)
            bn2 = P[b + 'bn2']
            check(lib.ha2g_se_bwd_scale_bn_f32(dout.data_ptr(), out.data_ptr(), c2.data_ptr(), m2.data_ptr(), s2.data_ptr(), bn2.gamma.data_ptr(),
                                               bn2.beta.data_ptr(), ds.data_ptr(), N, HW, C, sc.data_ptr(), ops.workspace(dout.device).data_ptr(), _stream()))
        else:
            check(lib.ha2g_se_bwd_scale_f32(dout.data_ptr(), out.data_ptr(), b2.data_ptr(), ds.data_ptr(), N, HW, C, sc.data_ptr(),
                                            ops.workspace(dout.device).data_ptr(), _stream()))
        dsc = ds                                            # already times the gate's sigmoid' (folded into the reduction's final pass)
        dh1, dpool = se_mlp_bwd(dsc, h1, w2_, w0_, HW)
    if not fused_tail:
        sink.gse(b, dsc, h1, dh1, pooled)
        dres, db2 = torch.empty_like(c2), torch.empty_like(c2)
        check(lib.ha2g_se_bwd_apply_f32(dout.data_ptr(), out.data_ptr(), sc.data_ptr(), dpool.data_ptr(), dres.data_ptr(),
                                        db2.data_ptr(), N, HW, C, _stream()))
        # the BatchNorm-backward apply pass is the PRODUCER of the convolutions' dy: where the plane-based data gradient serves the geometry it
        # also writes dy as bf16 hi / lo planes (same values the consumer tiles used to split out of the fp32 tensor, once instead of per tile)
        dc2 = sink.gbn(b + 'bn2', _rows(db2), _rows(c2), m2, s2, planes=p2, need_dx=f2)
        dc2, dc2p = ((dc2[0].view(c2.shape) if f2 else None), dc2[1]) if p2 else (dc2.view(c2.shape), None)
    # WGRAD_AFTER: the side stream's weight gradient is forked AFTER the data gradient of the same convolution has been enqueued, so that it
    # runs beside the bandwidth-bound BatchNorm-backward passes that follow instead of beside the (L2 -> LDS bound) data gradient
    if not WGRAD_AFTER:
        sink.gconv(b + 'conv2.weight', a1, dc2, wb, 1, 1, dy_planes=dc2p, x_planes=a1p)
    st1 = None
    nb1 = dgrad_bnstats_blocks(wb, c1.shape, 1, 1) if (p2 and dc2p.shape[0] == 3 and c1.dtype == torch.float32 and c1.is_contiguous()) else 0
    if nb1 > 0:
        # conv2's data gradient IS bn1's dy: its epilogue leaves bn1's backward sums behind, the column pass over (dy, c1) is not run
        da1, st1 = conv_dgrad_planes(dc2p, wb, a1.shape, 1, 1, bnstats=(c1, m1, s1, nb1))
    else:
        da1 = conv_dgrad_planes(dc2p, wb, a1.shape, 1, 1) if p2 else conv_dgrad(dc2, wb, a1.shape, 1, 1)
    if WGRAD_AFTER:
        sink.gconv(b + 'conv2.weight', a1, dc2, wb, 1, 1, dy_planes=dc2p, x_planes=a1p)
    if JOIN_DGRAD:
        ops.side.join(dx.device)
    dc1 = sink.gbn(b + 'bn1', _rows(da1), _rows(c1), m1, s1, relu_mask=True, planes=p1, need_dx=f1, partials=st1)
    dc1, dc1p = ((dc1[0].view(c1.shape) if f1 else None), dc1[1]) if p1 else (dc1.view(c1.shape), None)
    if not WGRAD_AFTER:
        sink.gconv(b + 'conv1.weight', x, dc1, wa, stride, 1, dy_planes=dc1p, x_planes=xp)
    if cd is None:                                                              # identity shortcut: accumulate onto d(residual)
        rs_ = resid if fused_tail else None
        if p1:
            r = conv_dgrad_planes(dc1p, wa, x.shape, stride, 1, out=dres, beta=1.0, resid=rs_)
        else:
            r = conv_dgrad(dc1, wa, x.shape, stride, 1, out=dres, beta=1.0, resid=rs_)
        if WGRAD_AFTER:
            sink.gconv(b + 'conv1.weight', x, dc1, wa, stride, 1, dy_planes=dc1p, x_planes=xp)
        if JOIN_DGRAD:
            ops.side.join(dx.device)
        return r
    dxin = conv_dgrad_planes(dc1p, wa, x.shape, stride, 1) if p1 else conv_dgrad(dc1, wa, x.shape, stride, 1)
    if WGRAD_AFTER:
        sink.gconv(b + 'conv1.weight', x, dc1, wa, stride, 1, dy_planes=dc1p, x_planes=xp)
    wd = _ohwi(P[b + 'downsample.0.weight'])
    pd = dgrad_planes_ok(wd, 2, 0)
    dcd = sink.gbn(b + 'downsample.1', _rows(dres), _rows(cd), md, sd, planes=pd)
    dcd, dcdp = (dcd[0].view(cd.shape), dcd[1]) if pd else (dcd.view(cd.shape), None)
    sink.gconv(b + 'downsample.0.weight', x, dcd, wd, 2, 0)
    if pd:
        conv_dgrad_planes(dcdp, wd, x.shape, 2, 0, out=dxin, beta=1.0)
    else:
        conv_dgrad(dcd, wd, x.shape, 2, 0, out=dxin, beta=1.0)
    if JOIN_DGRAD:
        ops.side.join(dx.device)
    return dxin


# ---- one tap (ResNetSE34V2.py:157-212): [PixelShuffle] -> conv -> ReLU -> BN -> flatten (B, W, C*H) -> FC -----------------

def tap_fwd(f, P, t, r):
    fin = _pixel_shuffle(f, r) if r > 1 else f
    wt = _ohwi(P['conv_%s.weight' % t])
    ct = conv_fwd(fin, wt, P['conv_%s.bias' % t], 1, 0, ACT_RELU)
    at, mt, st = _bn_fwd(ct, P['bn_%s' % t])
    packed = _tap_pack(at)                                                  # [B*Wt, C*Ht]
    y = ops.gemm(packed, P['fc_%s.weight' % t], transb=True, bias=P['fc_%s.bias' % t])
    return y.view(f.shape[0], at.shape[2], 32), (f.shape, fin, ct, mt, st, at.shape, packed)


def tap_bwd(dy, saved, P, t, r, sink):
    """dy [B, Wt, 32] -> gradient w.r.t. the trunk feature the tap reads (NHWC)."""
    fshape, fin, ct, mt, st, ashape, packed = saved
    dy = dy.view(fshape[0] * ashape[2], 32)
    sink.gwb('fc_%s.weight' % t, 'fc_%s.bias' % t, dy, packed)
    dpacked = ops.gemm(dy, P['fc_%s.weight' % t])
    dat = _tap_pack(dpacked, inverse=True, shape=ashape)
    dct = sink.gbn('bn_%s' % t, _rows(dat), _rows(ct), mt, st, relu_mask=True).view(ct.shape)     # BN' and ReLU' in one pass
    wt = _ohwi(P['conv_%s.weight' % t])
    sink.gconv('conv_%s.weight' % t, fin, dct, wt, 1, 0)
    sink.gb('conv_%s.bias' % t, _rows(dct))
    dfin = conv_dgrad(dct, wt, fin.shape, 1, 0)
    return _pixel_shuffle(dfin, r, inverse=True, shape=fshape) if r > 1 else dfin


# ---- speaker-conditioned softmax blending (ResNetSE34V2.py:196-216) ----------------------------------------------------------

def blend_fwd(vid, low, mid, high, P, L):
    B, T, _ = low.shape
    vid = vid.contiguous()
    ze = empty(B, 16, like=low)
    check(lib.ha2g_embedding_fwd_f32(vid.data_ptr(), P['speaker_embedding.0.weight'].data_ptr(), ze.data_ptr(), B, 16, _stream()))
    z = ops.gemm(ze, P['speaker_embedding.1.weight'], transb=True, bias=P['speaker_embedding.1.bias'])
    e0 = ops.eltwise(ops.OP_ELU, z)
    f1 = ops.gemm(e0, P['fc1.weight'], transb=True, bias=P['fc1.bias'])
    e1 = ops.eltwise(ops.OP_ELU, f1)
    logits = ops.gemm(e1, P['fc2.weight'], transb=True, bias=P['fc2.bias'])    # [B, 3*L] == (B,3,L)
    wsm = empty(B, 3, L, like=low)
    blend = empty(L, B, T, 32, like=low)
    check(lib.ha2g_blend_fwd_f32(logits.data_ptr(), low.data_ptr(), mid.data_ptr(), high.data_ptr(), wsm.data_ptr(), blend.data_ptr(),
                                 B, L, T * 32, _stream()))
    return wsm, blend, (vid, ze, e0, e1, wsm)


def blend_bwd(dw_ext, dblend, df, saved, feats, P, L, sink):
    """dblend: tuple of L gradients (or None) of the blended features, dw_ext: gradient of the softmax weights (or None);
    df = [dlow, dmid, dhigh] buffers the blend's contribution is ADDED to."""
    vid, ze, e0, e1, wsm = saved
    low, mid, high = feats
    B, T, _ = low.shape
    dev = low.device
    db = torch.stack([d if d is not None else torch.zeros_like(low) for d in dblend]).contiguous()
    dlogits = empty(B, 3 * L, like=low)
    dwe = dw_ext.contiguous() if dw_ext is not None else None
    check(lib.ha2g_blend_bwd_f32(db.data_ptr(), _p(dwe), wsm.data_ptr(), low.data_ptr(), mid.data_ptr(), high.data_ptr(),
                                 df[0].data_ptr(), df[1].data_ptr(), df[2].data_ptr(), dlogits.data_ptr(), B, L, T * 32, _stream()))
    # speaker MLP backward
    sink.gwb('fc2.weight', 'fc2.bias', dlogits, e1)
    de1 = ops.gemm(dlogits, P['fc2.weight'])
    df1 = ops.eltwise(ops.OP_ELU_BWD, de1, e1)
    sink.gwb('fc1.weight', 'fc1.bias', df1, e0)
    de0 = ops.gemm(df1, P['fc1.weight'])
    dz = ops.eltwise(ops.OP_ELU_BWD, de0, e0)
    sink.gwb('speaker_embedding.1.weight', 'speaker_embedding.1.bias', dz, ze)
    dze = ops.gemm(dz, P['speaker_embedding.1.weight'])
    demb = torch.zeros_like(P['speaker_embedding.0.weight'])
    check(lib.ha2g_embedding_bwd_f32(vid.data_ptr(), dze.data_ptr(), demb.data_ptr(), B, 16, -1, workspace(dev).data_ptr(), _stream()))
    sink.G['speaker_embedding.0.weight'] = demb


SAVED_TAP = [None]      # diagnostics: a list here receives the saved-activation dict S of every fp32-storage forward (relu_pattern_of reads it)


def block_relu_pattern(saved, prefix=''):
    """The ReLU decisions one block_fwd() took, from its saved tuple: {prefix + 'c1' | 'se' | 'out': bool tensor, NCHW / [N, R]} -- the key
    names of oracle.ha2g_oracle.relu_pattern (the test-side float64 oracle is linearised at this pattern)."""
    c1, h1, out = saved[1], saved[10], saved[16]
    return {prefix + 'c1': (c1 > 0).permute(0, 3, 1, 2), prefix + 'se': h1 > 0, prefix + 'out': (out > 0).permute(0, 3, 1, 2)}


def relu_pattern_of(S, prefix=''):
    """All ReLU decisions of one forward of the tower (fp32 storage): stem, every block, the three taps; keys as oracle.ha2g_oracle.relu_pattern
    with the blocks' keys prefixed by `prefix` (the oracle's state-dict prefix, e.g. 'audio.feat_extractor.')."""
    m = {'stem': (S['stem'][1] > 0).permute(0, 3, 1, 2)}
    for li, nblk in enumerate(LAYERS):
        for j in range(nblk):
            b = 'layer%d.%d.' % (li + 1, j)
            m.update(block_relu_pattern(S[b], prefix + b))
    for t, _, _, _ in TAPS:
        m['tap_' + t] = (S['tap_' + t][2] > 0).permute(0, 3, 1, 2)
    return m


class WavEncoderFunction(torch.autograd.Function):
    """apply(spec [B,128,W], vid [B], pose_level, names, bufs, *tensors) -> (weight, low, mid, high, blend_0..L-1).
    `tensors` follow param_names(): a BatchNorm entry contributes 2 tensors (gamma, beta); its buffers
    (running_mean, running_var, num_batches_tracked) come from the non-differentiable dict `bufs`."""

    @staticmethod
    def forward(ctx, spec, vid, L, names, bufs, *tensors):
        bufs, training = bufs
        if not training and torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
            raise NotImplementedError('ha2g_amd: the eval-mode audio encoder is forward-only; wrap inference in torch.no_grad()')
        _TRAINING[0] = training
        _NBT_PENDING.clear()
        P = {}
        it = iter(range(len(tensors)))
        flat_index = {}
        for n in names:
            if n.split('.')[-1].startswith('bn') or n.endswith('downsample.1'):
                idx = [next(it) for _ in range(2)]
                P[n] = _BN(tensors[idx[0]], tensors[idx[1]], *bufs[n])
                flat_index[n] = idx
            else:
                i = next(it)
                P[n] = tensors[i]
                flat_index[n] = i
        spec = spec.contiguous().float()
        B, H0, W0 = spec.shape
        S = {}                                            # saved activations for backward
        ctx.b16 = B16[0]
        if ctx.b16:
            from . import wav_b16
            feats16, S = wav_b16.trunk_fwd(spec, P, LAYERS, training, _NBT_PENDING)
            feats = [None] + [wav_b16.to_f32(f) for f in feats16[1:]]      # the taps read fp32
            return WavEncoderFunction._finish_forward(ctx, feats, S, P, vid, L, names, flat_index, tensors)
        # ---- stem: conv(1->32) + ReLU, then BN ----
        w1 = P['conv1.weight'].contiguous()
        c0 = empty(B, H0, W0, 32, like=spec)
        check(lib.ha2g_stem_conv_fwd_f32(spec.data_ptr(), w1.data_ptr(), P['conv1.bias'].data_ptr(), c0.data_ptr(), B, H0, W0, _stream()))
        x, m, s = _bn_fwd(c0, P['bn1'])
        S['stem'] = (spec, c0, m, s)
        feats = []
        # a backward will follow iff some input needs a gradient (false under no_grad); train / eval mode only governs the BatchNorm statistics
        _WILL_BWD[0] = bool(any(ctx.needs_input_grad))
        _FWD_PLANES[0] = (PLANES & 6) == 6 and _WILL_BWD[0]                                # false under no_grad: nothing will read the planes
        try:                                                # the two flags are module globals: restored whatever block_fwd raises (OOM, check())
            wpl = prepare_fwd_weight_planes(P)                  # {} unless the forward of layers 2-4 runs on three-piece planes (fp32-class default mode)
            blocks = [('layer%d.%d.' % (li + 1, j), li, j) for li, nblk in enumerate(LAYERS) for j in range(nblk)]
            xp = None
            for bi, (b, li, j) in enumerate(blocks):
                # pieces the block's output is written with: 3 when the NEXT block's convolutions of it run on planes, else the round-3 opt-in (2)
                nxt = blocks[bi + 1][0] if bi + 1 < len(blocks) else None
                want = 0
                if nxt is not None and (nxt + 'conv1.weight') in wpl:
                    want = 3
                elif nxt is not None and blocks[bi + 1][1] == li and _FWD_PLANES[0]:
                    # the reader is the NEXT block's conv1 weight gradient (stride 1, same layer): judge ITS geometry -- this block's OUTPUT, which is
                    # half the input's size when this block is the layer's stride-2 block
                    oh, ow = ((x.shape[1] + 1) // 2, (x.shape[2] + 1) // 2) if (j == 0 and li > 0) else (x.shape[1], x.shape[2])
                    if wgrad_planes_ok(None, _ohwi(P[nxt + 'conv1.weight']), 1, 1, hw=(oh, ow)):
                        want = 2
                x, S[b], xp = block_fwd(x, P, b, j == 0 and li > 0, xp=xp, out_planes=want, wpl=wpl)
                if j + 1 == LAYERS[li]:
                    feats.append(x)
        finally:
            _FWD_PLANES[0], _WILL_BWD[0] = False, True
        return WavEncoderFunction._finish_forward(ctx, feats, S, P, vid, L, names, flat_index, tensors)

    @staticmethod
    def _finish_forward(ctx, feats, S, P, vid, L, names, flat_index, tensors):
        # ---- taps ----
        tap_out = []
        for (t, C, k, r), f in zip(TAPS, feats[1:]):
            y, S['tap_' + t] = tap_fwd(f, P, t, r)
            tap_out.append(y)
        low, mid, high = tap_out
        if _NBT_PENDING:
            torch._foreach_add_(_NBT_PENDING, 1)
            _NBT_PENDING.clear()
        wsm, blend, S['spk'] = blend_fwd(vid, low, mid, high, P, L)
        S['feats'] = (low, mid, high)
        if SAVED_TAP[0] is not None and not ctx.b16:
            SAVED_TAP[0].append(S)
        ctx.S, ctx.P, ctx.L, ctx.names, ctx.flat_index, ctx.n_tensors = S, P, L, names, flat_index, len(tensors)
        return (wsm, low, mid, high) + tuple(blend[i] for i in range(L))

    @staticmethod
    def backward(ctx, dw_ext, dlow, dmid, dhigh, *dblend):
        try:
            return WavEncoderFunction._backward(ctx, dw_ext, dlow, dmid, dhigh, *dblend)
        finally:
            _WPLANES[0] = None                           # a failed backward must not leave another step's weight planes behind

    @staticmethod
    def _backward(ctx, dw_ext, dlow, dmid, dhigh, *dblend):
        S, P, L = ctx.S, ctx.P, ctx.L
        sink = GradSink(P)
        G = sink.G
        low, mid, high = S['feats']
        dev = low.device

        def z_or(t, like):
            return t.contiguous().clone() if t is not None else torch.zeros_like(like)

        df = [z_or(dlow, low), z_or(dmid, mid), z_or(dhigh, high)]
        if any(d is not None for d in dblend) or dw_ext is not None:
            blend_bwd(dw_ext, dblend, df, S['spk'], S['feats'], P, L, sink)
        # ---- taps backward -> gradient w.r.t. the three trunk features ----
        dfeat = [None, None, None, None]
        for ti, (t, C, k, r) in enumerate(TAPS):
            dfeat[ti + 1] = tap_bwd(df[ti], S['tap_' + t], P, t, r, sink)
        # ---- trunk backward ----
        if WPLANES_MULTI and not ctx.b16:
            # keyed by the address of the weight's physical OHWI image -- only where that image is a VIEW of the parameter (channels_last storage):
            # a `.contiguous()` temporary is freed at once and the caching allocator may hand its address to the next weight of the same shape
            _WPLANES[0] = {P[n].data_ptr(): v for n, v in prepare_weight_planes(P).items() if P[n].permute(0, 2, 3, 1).is_contiguous()}
        if ctx.b16:
            from . import wav_b16
            wav_b16.trunk_bwd(dfeat, S, P, LAYERS, sink)
            return WavEncoderFunction._finish_backward(ctx, sink, dev)
        dx = None
        for li in range(len(LAYERS) - 1, -1, -1):
            if dfeat[li] is not None:
                dx = dfeat[li] if dx is None else ops.eltwise(ops.OP_ADD, dx, dfeat[li])
            for j in range(LAYERS[li] - 1, -1, -1):
                b = 'layer%d.%d.' % (li + 1, j)
                dx = block_bwd(dx, S[b], P, b, sink)
        # ---- stem backward ----
        spec, c0, m0, s0 = S['stem']
        dc0 = sink.gbn('bn1', _rows(dx), _rows(c0), m0, s0, relu_mask=True)
        dw1, dbias1 = torch.empty_like(P['conv1.weight'].contiguous()), torch.empty_like(P['conv1.bias'])
        Bn, H0, W0 = spec.shape
        check(lib.ha2g_stem_conv_wgrad_f32(spec.data_ptr(), dc0.data_ptr(), dw1.data_ptr(), dbias1.data_ptr(), Bn, H0, W0, 0.0,
                                           workspace(dev).data_ptr(), _stream()))
        G['conv1.weight'], G['conv1.bias'] = dw1, dbias1
        return WavEncoderFunction._finish_backward(ctx, sink, dev)

    @staticmethod
    def _finish_backward(ctx, sink, dev):
        _WPLANES[0] = None
        G = sink.G
        sink.join(dev)                                   # the side stream's weight gradients are complete before autograd sees them
        # ---- scatter into the flat gradient tuple ----
        grads = [None] * ctx.n_tensors
        for n, idx in ctx.flat_index.items():
            if n not in G:
                continue
            if isinstance(idx, list):
                grads[idx[0]], grads[idx[1]] = G[n]
            else:
                grads[idx] = G[n]
        ctx.S = None
        return (None, None, None, None, None) + tuple(grads)

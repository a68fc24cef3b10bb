"""bf16-STORAGE trunk of the SE-ResNet34 audio encoder (BASELINE config 5, `bench.py --bf16`; opt-in, never the default path).

Every activation of the trunk (stem output, the five tensors of each SEBasicBlock, the residual stream) and every activation gradient of its
backward live in HBM as bf16; BatchNorm statistics, the SE squeeze / excitation, gamma / beta / weight gradients and all accumulators stay
fp32 (double inside the column reductions), convolutions accumulate in fp32 on v_mfma_f32_32x32x16_bf16 with bf16-rounded weights.  The
bandwidth-bound passes (BatchNorm, SE, residual) therefore move half the bytes of the fp32 mode, and the matrix kernels read single bf16
planes (one MFMA per product instead of the split's three).  The three taps, the speaker blend and everything downstream stay fp32: the
trunk features are widened where the taps read them and the taps' gradients are rounded where they join the trunk's gradient stream.

Reference being replaced: scripts/model/ResNetSE34V2.py:118-156 and scripts/model/ResNetBlocks.py:21-37,81-95 under autograd
(train_eval/train_hierarchy.py:264) -- the reference itself has no reduced-precision mode; tests/test_gpu_b16.py holds this path against the
oracle evaluated with the same rounding points.
"""
import torch

from . import ops
from ._lib import check, lib
from .ops import ACT_NONE, _p, _stream, workspace

BF = torch.bfloat16


def e16(*shape, device):
    return torch.empty(shape, dtype=BF, device=device)


def to_b16(x):
    """fp32 -> bf16, round to nearest even (numel % 4 == 0)."""
    y = torch.empty(x.shape, dtype=BF, device=x.device)
    check(lib.ha2g_f32_to_b16(x.data_ptr(), y.data_ptr(), x.numel(), _stream()))
    return y


def to_f32(x):
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(lib.ha2g_b16_to_f32(x.data_ptr(), y.data_ptr(), x.numel(), _stream()))
    return y


def add_into(a16, b32):
    """bf16(a16 + b32); a16 None: bf16(b32)."""
    out = torch.empty(b32.shape, dtype=BF, device=b32.device)
    check(lib.ha2g_add_f32_to_b16(_p(a16), b32.data_ptr(), out.data_ptr(), b32.numel(), _stream()))
    return out


def wt_b16(w_ohwi):
    """fp32 [Cout][KH][KW][Cin] -> bf16 [Cin][KH][KW][Cout] (the data gradient's B operand)."""
    Cout, KH, KW, Cin = w_ohwi.shape
    wh = torch.empty(Cin, KH, KW, Cout, dtype=BF, device=w_ohwi.device)
    wl = torch.empty_like(wh)
    check(lib.ha2g_conv2d_weight_ihwo_planes(w_ohwi.data_ptr(), wh.data_ptr(), wl.data_ptr(), Cout, KH, KW, Cin, _stream()))
    return wh


def conv_fwd(x, w16, stride, pad, relu):
    N, H, W, Cin = x.shape
    Cout, KH, KW, _ = w16.shape
    OH, OW = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    y = e16(N, OH, OW, Cout, device=x.device)
    ops.ktimer.launch('conv2d_fwd_b16', lambda: check(lib.ha2g_conv2d_fwd_b16(
        x.data_ptr(), w16.data_ptr(), y.data_ptr(), N, H, W, Cin, Cout, KH, KW, stride, pad, int(relu), _stream())),
        2.0 * N * OH * OW * Cout * KH * KW * Cin)
    return y


def conv_dgrad(dy, w_ohwi, xshape, stride, pad, out=None, beta=0.0):
    N, H, W, Cin = xshape
    Cout, KH, KW, _ = w_ohwi.shape
    wt = wt_b16(w_ohwi)
    if out is None:
        out = e16(N, H, W, Cin, device=dy.device)
        beta = 0.0
    if beta == 0.0 and stride == 2 and KH == 1:
        out.zero_()
    ops.ktimer.launch('conv_dgrad_b16', lambda: check(lib.ha2g_conv2d_dgrad_b16(
        dy.data_ptr(), wt.data_ptr(), out.data_ptr(), N, H, W, Cin, Cout, KH, KW, stride, pad, beta, _stream())),
        2.0 * N * H * W * Cin * KH * KW * Cout)
    return out


def gconv(sink, name, x, dy, w_ohwi, stride, pad):
    """Weight gradient of one convolution from bf16 x / dy, on the side stream like the fp32 mode's: the single-plane kernels where they
    serve the geometry (3x3 stride 1: the DMA-staged plane kernel for layers 2-4, the direct 32-channel kernel for layer 1), else the fp32
    kernels on widened copies (the stride-2 and 1x1 convolutions of a layer's first block: three small tensors per layer)."""
    from . import wav_engine as we
    N, H, W, Cin = x.shape
    Cout, KH, KW, _ = w_ohwi.shape
    dev = x.device
    side_on = we.SIDE_WGRAD and ops.side.enabled and x.is_cuda
    with (ops.side.section(dev) if side_on else ops._null()):
        if side_on:
            st = ops.cur_stream(dev)
            x.record_stream(st); dy.record_stream(st)
        into = sink.tgt(sink.P[name])
        if lib.ha2g_conv2d_wgrad_b16_supported(H, W, Cin, Cout, KH, KW, stride, pad):
            beta = 0.0
            if into is not None:
                dwp = into.permute(0, 2, 3, 1)
                if dwp.is_contiguous():
                    dw, beta = dwp, 1.0
                else:
                    into = None
            if into is None:
                dw = torch.empty(Cout, KH, KW, Cin, dtype=torch.float32, device=dev)
            ws = workspace(dev)
            assert lib.ha2g_conv2d_wgrad_b16_workspace_bytes(N, H, W, Cin, Cout) <= ws.numel() * 4
            ops.ktimer.launch('conv_wgrad_b16', lambda: check(lib.ha2g_conv2d_wgrad_b16(
                x.data_ptr(), dy.data_ptr(), dw.data_ptr(), N, H, W, Cin, Cout, KH, KW, stride, pad, beta, ws.data_ptr(), ws.numel() * 4,
                _stream())), 2.0 * N * H * W * Cin * KH * KW * Cout)
            r = None if into is not None else dw.permute(0, 3, 1, 2)
        else:
            r = we.conv_wgrad(to_f32(x), to_f32(dy), w_ohwi, stride, pad, into=into)
        if r is not None and side_on:
            r.record_stream(torch.cuda.default_stream(dev))
    if side_on:
        sink.forked = True
    if r is not None:
        sink.G[name] = r


def bn_fwd(x, bn, training, nbt_pending, pool=False):
    """x bf16 [N,H,W,C] -> (y bf16, mean, invstd[, pooled fp32 [N,C]])"""
    N, H, W, C = x.shape
    rows = N * H * W
    dev = x.device
    if not training:
        mean, invstd = bn.rm, ops.eltwise(ops.OP_RSQRT_EPS, bn.rv, alpha=1e-5)
    else:
        mean, invstd = torch.empty(C, dtype=torch.float32, device=dev), torch.empty(C, dtype=torch.float32, device=dev)
        ops.ktimer.launch('bn_stats_b16', lambda: check(lib.ha2g_bn_stats_b16(
            x.data_ptr(), rows, C, mean.data_ptr(), invstd.data_ptr(), _p(bn.rm), _p(bn.rv), 0.1, 1e-5, workspace(dev).data_ptr(), _stream())),
            2.0 * rows * C)
        if bn.nbt is not None:
            nbt_pending.append(bn.nbt)
    y = torch.empty_like(x)
    if pool:
        pooled = torch.empty(N, C, dtype=torch.float32, device=dev)
        ws = workspace(dev)
        assert lib.ha2g_bn_apply_pool_workspace_floats(N, H * W, C) <= ws.numel()
        check(lib.ha2g_bn_apply_pool_b16(x.data_ptr(), mean.data_ptr(), invstd.data_ptr(), bn.gamma.data_ptr(), bn.beta.data_ptr(), y.data_ptr(),
                                         N, H * W, C, pooled.data_ptr(), ws.data_ptr(), _stream()))
        return y, mean, invstd, pooled
    check(lib.ha2g_bn_apply_b16(x.data_ptr(), mean.data_ptr(), invstd.data_ptr(), bn.gamma.data_ptr(), bn.beta.data_ptr(), y.data_ptr(), rows, C,
                                ACT_NONE, _stream()))
    return y, mean, invstd


def gbn(sink, name, dy, x, mean, invstd, relu_mask=False):
    """BatchNorm backward on bf16 dy / x -> dx bf16 (gamma / beta gradients fp32, to the sink)."""
    bn = sink.P[name]
    C = x.shape[-1]
    rows = x.numel() // C
    dev = x.device
    tg_, tb_ = sink.tgt(bn.gamma), sink.tgt(bn.beta)
    acc = (tg_, tb_) if (tg_ is not None and tb_ is not None) else (None, None)
    dx = torch.empty_like(x)
    dgamma, dbeta = torch.empty(C, dtype=torch.float32, device=dev), torch.empty(C, dtype=torch.float32, device=dev)
    ops.ktimer.launch('bn_bwd_b16', lambda: check(lib.ha2g_bn_bwd_b16(
        dy.data_ptr(), x.data_ptr(), mean.data_ptr(), invstd.data_ptr(), bn.gamma.data_ptr(), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
        rows, C, int(relu_mask), _p(acc[0]), _p(acc[1]), workspace(dev).data_ptr(), _stream())), 2.0 * rows * C * 5)
    if acc[0] is None:
        sink.G[name] = (dgamma, dbeta)
    return dx


def block_fwd(x, P, b, first, training, nbt):
    """x bf16 NHWC -> (out bf16, saved)"""
    from .wav_engine import _ohwi
    stride = 2 if first else 1
    wa, wb = _ohwi(P[b + 'conv1.weight']), _ohwi(P[b + 'conv2.weight'])
    c1 = conv_fwd(x, to_b16(wa), stride, 1, True)
    a1, m1, s1 = bn_fwd(c1, P[b + 'bn1'], training, nbt)
    c2 = conv_fwd(a1, to_b16(wb), 1, 1, False)
    b2, m2, s2, pooled = bn_fwd(c2, P[b + 'bn2'], training, nbt, pool=True)
    N, OH, OW, C = b2.shape
    h1 = ops.gemm(pooled, P[b + 'se.fc.0.weight'], transb=True, bias=P[b + 'se.fc.0.bias'], act=ops.ACT_RELU)
    sc = ops.gemm(h1, P[b + 'se.fc.2.weight'], transb=True, bias=P[b + 'se.fc.2.bias'], act=ops.ACT_SIGMOID)
    if first:
        wd = _ohwi(P[b + 'downsample.0.weight'])
        cd = conv_fwd(x, to_b16(wd), 2, 0, False)
        res, md, sd = bn_fwd(cd, P[b + 'downsample.1'], training, nbt)
    else:
        res, cd, md, sd = x, None, None, None
    out = torch.empty_like(b2)
    check(lib.ha2g_se_scale_add_relu_b16(b2.data_ptr(), sc.data_ptr(), res.data_ptr(), out.data_ptr(), N, OH * OW, C, _stream()))
    return out, (x, c1, m1, s1, a1, c2, m2, s2, b2, pooled, h1, sc, cd, md, sd, out, stride)


def block_bwd(dout, saved, P, b, sink):
    """dout = d(out) bf16 NHWC -> d(x) bf16 NHWC"""
    from .wav_engine import _ohwi
    (x, c1, m1, s1, a1, c2, m2, s2, b2, pooled, h1, sc, cd, md, sd, out, stride) = saved
    N, OH, OW, C = b2.shape
    HW = OH * OW
    dev = dout.device
    ds = torch.empty(N, C, dtype=torch.float32, device=dev)
    check(lib.ha2g_se_bwd_scale_b16(dout.data_ptr(), out.data_ptr(), b2.data_ptr(), ds.data_ptr(), N, HW, C, sc.data_ptr(),
                                    workspace(dev).data_ptr(), _stream()))
    sink.gwb(b + 'se.fc.2.weight', b + 'se.fc.2.bias', ds, h1)
    from .wav_engine import se_mlp_bwd
    dh1, dpool = se_mlp_bwd(ds, h1, P[b + 'se.fc.2.weight'], P[b + 'se.fc.0.weight'], HW)
    sink.gwb(b + 'se.fc.0.weight', b + 'se.fc.0.bias', dh1, pooled)
    dres, db2 = torch.empty_like(b2), torch.empty_like(b2)
    check(lib.ha2g_se_bwd_apply_b16(dout.data_ptr(), out.data_ptr(), sc.data_ptr(), dpool.data_ptr(), dres.data_ptr(), db2.data_ptr(), N, HW, C,
                                    _stream()))
    wb, wa = _ohwi(P[b + 'conv2.weight']), _ohwi(P[b + 'conv1.weight'])
    dc2 = gbn(sink, b + 'bn2', db2, c2, m2, s2)
    gconv(sink, b + 'conv2.weight', a1, dc2, wb, 1, 1)
    da1 = conv_dgrad(dc2, wb, a1.shape, 1, 1)
    dc1 = gbn(sink, b + 'bn1', da1, c1, m1, s1, relu_mask=True)
    gconv(sink, b + 'conv1.weight', x, dc1, wa, stride, 1)
    if cd is None:
        return conv_dgrad(dc1, wa, x.shape, stride, 1, out=dres, beta=1.0)
    dxin = conv_dgrad(dc1, wa, x.shape, stride, 1)
    wd = _ohwi(P[b + 'downsample.0.weight'])
    dcd = gbn(sink, b + 'downsample.1', dres, cd, md, sd)
    gconv(sink, b + 'downsample.0.weight', x, dcd, wd, 2, 0)
    return conv_dgrad(dcd, wd, x.shape, 2, 0, out=dxin, beta=1.0)


def trunk_fwd(spec, P, layers, training, nbt):
    """spec fp32 [B,H0,W0] -> (feats: the four layer outputs, bf16 NHWC; S: saved activations)"""
    B, H0, W0 = spec.shape
    S = {}
    w1 = P['conv1.weight'].contiguous()
    c0 = e16(B, H0, W0, 32, device=spec.device)
    check(lib.ha2g_stem_conv_fwd_b16(spec.data_ptr(), w1.data_ptr(), P['conv1.bias'].data_ptr(), c0.data_ptr(), B, H0, W0, _stream()))
    x, m, s = bn_fwd(c0, P['bn1'], training, nbt)
    S['stem'] = (spec, c0, m, s)
    feats = []
    for li, nblk in enumerate(layers):
        for j in range(nblk):
            b = 'layer%d.%d.' % (li + 1, j)
            x, S[b] = block_fwd(x, P, b, j == 0 and li > 0, training, nbt)
        feats.append(x)
    return feats, S


def trunk_bwd(dfeat, S, P, layers, sink):
    """dfeat[li]: fp32 gradient of layer li's output from its tap (or None); parameter gradients to the sink."""
    dx = None
    for li in range(len(layers) - 1, -1, -1):
        if dfeat[li] is not None:
            dx = add_into(dx, dfeat[li])
        for j in range(layers[li] - 1, -1, -1):
            b = 'layer%d.%d.' % (li + 1, j)
            dx = block_bwd(dx, S[b], P, b, sink)
    spec, c0, m0, s0 = S['stem']
    dc0 = gbn(sink, 'bn1', dx, c0, m0, s0, relu_mask=True)
    dw1, dbias1 = torch.empty_like(P['conv1.weight'].contiguous()), torch.empty_like(P['conv1.bias'])
    Bn, H0, W0 = spec.shape
    check(lib.ha2g_stem_conv_wgrad_b16(spec.data_ptr(), dc0.data_ptr(), dw1.data_ptr(), dbias1.data_ptr(), Bn, H0, W0, 0.0,
                                       workspace(spec.device).data_ptr(), _stream()))
    sink.G['conv1.weight'], sink.G['conv1.bias'] = dw1, dbias1

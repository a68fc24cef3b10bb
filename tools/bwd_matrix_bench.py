"""Backward matrix kernels of the train step in isolation (B = 128): the SE-ResNet's convolution data / weight gradients per trunk shape and
the dominant dense backward GEMMs (GRU input-projection dW / dX), in the default three-piece (fp32-class) mode 70, the two-piece mode 6 and on the exact fp32 MFMA (mode 0).
Prints per kernel: microseconds, fp32-equivalent TFLOP/s, fraction of the 3-product split-bf16 roofline (2.5 PF / 3 = 833 TF).
Also the target of tools/pmc_bwd.sh (rocprofv3 PMC passes).   usage: python tools/bwd_matrix_bench.py [iters]"""
import sys
import torch
sys.path.insert(0, '.')
from ha2g_amd import ops, wav_engine as we
from ha2g_amd._lib import lib

dev = torch.device('cuda:0')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def timeit(fn, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


B = 128
rows = []
for mode in (70, 6, 0):
    lib.ha2g_gemm_set_mode(mode)
    for H, W, C in ((128, 70, 32), (64, 35, 64), (32, 18, 128), (16, 9, 256)):
        x = torch.randn(B, H, W, C, device=dev)
        w = torch.randn(C, 3, 3, C, device=dev) * 0.05
        dy = torch.randn(B, H, W, C, device=dev)
        fl = 2.0 * B * H * W * C * C * 9
        d = timeit(lambda: we.conv_dgrad(dy, w, x.shape, 1, 1))
        g = timeit(lambda: we.conv_wgrad(x, dy, w, 1, 1))
        rows.append((mode, 'conv3x3 dgrad C=%d %dx%d' % (C, H, W), d, fl))
        rows.append((mode, 'conv3x3 wgrad C=%d %dx%d' % (C, H, W), g, fl))
        if mode and C >= 64:                                      # the plane kernels on pre-split operands (what the train step runs on layers 2-4)
            dyp, xp = ops.to_planes(dy), ops.to_planes(x)
            rows.append((mode, 'conv3x3 dgrad C=%d %dx%d PLANES (producer-split)' % (C, H, W), timeit(lambda: we.conv_dgrad_planes(dyp, w, x.shape, 1, 1)), fl))
            rows.append((mode, 'conv3x3 wgrad C=%d %dx%d PLANES (producer-split)' % (C, H, W), timeit(lambda: we.conv_wgrad_planes(xp, dyp, w, x.shape)), fl))
            if mode == 70:
                lib.ha2g_conv_planes_tile3(6)
                rows.append((mode, 'conv3x3 dgrad C=%d %dx%d PLANES 32x32 kernels (q kernel off)' % (C, H, W), timeit(lambda: we.conv_dgrad_planes(dyp, w, x.shape, 1, 1)), fl))
                lib.ha2g_conv_planes_tile3(0)
                xp3, wp3 = ops.to_planes(x, 3), ops.to_planes(w, 3)
                rows.append((mode, 'conv3x3 FORWARD C=%d %dx%d PLANES' % (C, H, W), timeit(lambda: we.conv_fwd_planes(xp3, wp3, x.shape, 1, 1, 0)), fl))
            if mode == 0:
                rows.append((mode, 'conv3x3 FORWARD C=%d %dx%d fp32 implicit GEMM' % (C, H, W), timeit(lambda: we.conv_fwd(x, w, None, 1, 1, 0)), fl))
    for (M, N, K, name) in ((4352, 900, 600, 'GRU layer 1-3'), (13056, 900, 600, 'GRU fused 3-chain rows (fwd shape)')):
        dgi = torch.randn(M, N, device=dev)
        xin = torch.randn(M, K, device=dev)
        wih = torch.randn(N, K, device=dev) * 0.05
        fl = 2.0 * M * N * K
        rows.append((mode, 'dense dW = dgi^T x  [%d x %d x %d] %s' % (N, K, M, name), timeit(lambda: ops.gemm(dgi, xin, transa=True)), fl))
        rows.append((mode, 'dense dX = dgi W    [%d x %d x %d] %s' % (M, K, N, name), timeit(lambda: ops.gemm(dgi, wih)), fl))
from ha2g_amd._lib import DEFAULT_GEMM_MODE
lib.ha2g_gemm_set_mode(DEFAULT_GEMM_MODE)
print('%-4s %-66s %9s %9s %7s' % ('mode', 'kernel', 'us', 'TF(f32eq)', 'frac'))
for mode, name, us, fl in rows:
    tf = fl / us / 1e6
    peak = {70: 416.7, 6: 833.3, 0: 157.3}[mode]                 # bf16 dense peak / MFMAs per fp32-equivalent product (6 / 3), or the fp32 MFMA peak
    print('%-4d %-66s %9.1f %9.1f %7.3f' % (mode, name, us, tf, tf / peak))

"""The dense plane GEMM (ops.gemm routing, csrc/conv_planes.hip pconv_q_kernel) against the round-3 dense kernels on the step's dense shapes:
forward y = x W^T (fp32 MFMA GEMM before), dX = dY W and dW = dY^T X (three-piece in-kernel-split GEMM before); the plane numbers INCLUDE the
split / transpose passes of both operands.  usage: python tools/plane_gemm_bench.py"""
import sys
import torch
sys.path.insert(0, '.')
from ha2g_amd import ops
from ha2g_amd._lib import lib

dev = torch.device('cuda:0')


def t_us(fn, iters=20):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


shapes = [('fwd gi  x W^T', 13056, 900, 600, False, True), ('fwd gi0 x W^T', 13056, 900, 108, False, True), ('fwd gi (B=128 rows)', 4352, 900, 600, False, True),
          ('dX = dgi W', 4352, 600, 900, False, False), ('dW = dgi^T x', 900, 600, 4352, True, False), ('head 300->150', 4352, 150, 300, False, True),
          ('tcn 600->300', 13056, 300, 600, False, True), ('fwd gi both dirs N=1800', 13056, 1800, 600, False, True)]
print('%-28s %6s %6s %6s %10s %10s %10s %9s' % ('shape', 'M', 'N', 'K', 'old us', 'planes us', 'kernel us', 'TF(f32eq)'))
for name, M, N, K, ta, tb in shapes:
    a = torch.randn((K, M) if ta else (M, K), device=dev)
    b = torch.randn((N, K) if tb else (K, N), device=dev) * 0.05
    ops.PLANE_GEMM = False
    t_old = t_us(lambda: ops.gemm(a, b, transa=ta, transb=tb))
    ops.PLANE_GEMM = True
    old_min = ops.PLANE_GEMM_MIN_FLOP
    ops.PLANE_GEMM_MIN_FLOP = 0.0
    t_new = t_us(lambda: ops.gemm(a, b, transa=ta, transb=tb))
    ap, bp = ops._planes_2d(a, ta), ops._planes_2d(b, not tb)
    out = torch.empty(M, N, device=dev)
    ws = ops.workspace(dev)
    t_k = t_us(lambda: lib.ha2g_gemm_planes_np_f32(ap.data_ptr(), ap.stride(0), ap.shape[2], bp.data_ptr(), bp.stride(0), bp.shape[2], 3, M, N, K, 0.0,
                                                   out.data_ptr(), out.stride(0), None, 0, ws.data_ptr(), ws.numel() * 4, torch.cuda.current_stream().cuda_stream))
    ops.PLANE_GEMM_MIN_FLOP = old_min
    print('%-28s %6d %6d %6d %10.1f %10.1f %10.1f %9.1f' % (name, M, N, K, t_old, t_new, t_k, 2.0 * M * N * K / t_k / 1e6))

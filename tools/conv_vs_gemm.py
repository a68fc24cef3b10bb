"""How much of the implicit-GEMM convolution's time is the im2col gather: conv_fwd vs a plain NT GEMM of the same M, N, K."""
import sys, torch
sys.path.insert(0, '.')
from ha2g_amd import ops, wav_engine as we
from ha2g_amd._lib import lib
dev = torch.device('cuda:0')
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
B = 128
for H, W, C in ((128, 70, 32), (64, 35, 64), (32, 18, 128), (16, 9, 256)):
    x = torch.randn(B, H, W, C, device=dev); w = torch.randn(C, 3, 3, C, device=dev) * 0.05
    M, N, K = B * H * W, C, 9 * C
    a, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    fl = 2.0 * M * N * K
    tc = timeit(lambda: we.conv_fwd(x, w, None, 1, 1, 0))
    tg = timeit(lambda: ops.gemm(a, b, transb=True))
    print('C=%3d M=%7d N=%3d K=%4d: conv %.0f us %.0f TF | plain GEMM %.0f us %.0f TF' % (C, M, N, K, tc, fl / tc / 1e6, tg, fl / tg / 1e6))

"""Micro-benchmarks of individual kernels on one MI355X (HIP-event timed)."""
import sys
import torch
sys.path.insert(0, '.')
from ha2g_amd import ops
from ha2g_amd._lib import lib, check

dev = torch.device('cuda:0')


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def bench_gemm(M, N, K, ta=False, tb=True):
    a = torch.randn((K, M) if ta else (M, K), device=dev)
    b = torch.randn((N, K) if tb else (K, N), device=dev)
    c = torch.empty(M, N, device=dev)
    us = timeit(lambda: ops.gemm(a, b, ta, tb, out=c))
    print('gemm M=%d N=%d K=%d ta=%d tb=%d: %.1f us  %.1f TFLOP/s' % (M, N, K, ta, tb, us, 2.0 * M * N * K / us / 1e6))


def bench_gru(B, T=34, H=300):
    st = torch.cuda.current_stream().cuda_stream
    gi = torch.randn(B * T, 6 * H, device=dev)
    whh = torch.randn(3 * H, H, device=dev) / H ** 0.5
    npk = lib.ha2g_gru_packed_floats(H)
    pk = torch.empty(4, npk, device=dev)
    check(lib.ha2g_gru_pack_whh(whh.data_ptr(), pk[0].data_ptr(), pk[2].data_ptr(), H, st))
    check(lib.ha2g_gru_pack_whh(whh.data_ptr(), pk[1].data_ptr(), pk[3].data_ptr(), H, st))
    bhh = torch.randn(3 * H, device=dev)
    y = torch.empty(B, T, 2 * H, device=dev)
    rs = torch.empty(B, T, 2, 4, H, device=dev)
    dg = torch.empty(B * T, 8 * H, device=dev)
    dy = torch.randn(B, T, 2 * H, device=dev)
    f = timeit(lambda: check(lib.ha2g_gru_layer_fwd(gi.data_ptr(), pk.data_ptr(), bhh.data_ptr(), bhh.data_ptr(), y.data_ptr(),
                                                    rs.data_ptr(), B, T, H, st)))
    b = timeit(lambda: check(lib.ha2g_gru_layer_bwd(dy.data_ptr(), y.data_ptr(), rs.data_ptr(), pk[2].data_ptr(), dg.data_ptr(),
                                                    0, B, T, H, st)))
    fc = None
    if lib.ha2g_gru_cluster_supported(H):
        xch, err = ops._cluster_scratch(dev)
        fc = timeit(lambda: check(lib.ha2g_gru_layer_fwd_cluster(gi.data_ptr(), pk.data_ptr(), bhh.data_ptr(), bhh.data_ptr(), y.data_ptr(),
                                                                 rs.data_ptr(), xch.data_ptr(), err.data_ptr(), B, T, H, st)))
    fl = 2.0 * B * T * 2 * 3 * H * H
    print('gru layer B=%d T=%d H=%d: fwd %.1f us (%.2f us/step, %.1f TFLOP/s)  bwd %.1f us (%.1f TFLOP/s)' % (
        B, T, H, f, f / T, fl / f / 1e6, b, fl / b / 1e6) + ('' if fc is None else '  | cluster fwd %.1f us (%.2f us/step, %.1f TFLOP/s) err=%d' % (fc, fc / T, fl / fc / 1e6, ops.gru_cluster_error(dev))))


if __name__ == '__main__':
    for B in (16, 128, 256, 384):
        bench_gru(B)
    bench_gru(128, 28, 64)
    bench_gemm(4352, 1800, 600)
    bench_gemm(4352, 900, 108)
    bench_gemm(13056, 900, 600)
    bench_gemm(4352, 600, 900, False, False)
    bench_gemm(900, 600, 4352, True, False)
    bench_gemm(8192, 8192, 1024)
    bench_gemm(4096, 4096, 4096)
    bench_gemm(1146880, 32, 288)
    bench_gemm(286720, 64, 576)

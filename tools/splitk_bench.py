"""Split-K products alone: the two-launch form (kernel + splitk_reduce[_wide]_kernel, round 5) against the in-kernel last-arriver reduction (round 6),
and the plane GEMM's k-slice count on the GRU dX shape.  python tools/splitk_bench.py -> one line per shape (us per call, TFLOP/s)."""
import sys
import torch
sys.path.insert(0, '.')
from ha2g_amd import ops  # noqa: E402
from ha2g_amd._lib import lib  # noqa: E402

dev = torch.device('cuda:0')
ops.workspace(dev)


def t_us(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1000.0 / n


def dense(G, M, N, K, ta, tb):
    A = [torch.randn((K, M) if ta else (M, K), device=dev) for _ in range(G)]
    B = [torch.randn((N, K) if tb else (K, N), device=dev) for _ in range(G)]
    C = [torch.zeros(M, N, device=dev) for _ in range(G)]
    cs = [torch.zeros(M, device=dev) for _ in range(G)] if (ta and not tb) else None
    if G == 1:
        kw = dict(colsum_out=cs[0], colsum_beta=1.0) if cs is not None else {}
        return lambda: ops.gemm(A[0], B[0], transa=ta, transb=tb, out=C[0], beta=1.0, **kw)
    return lambda: ops.gemm_grouped(A, B, transa=ta, transb=tb, out=C, beta=1.0, colsum_out=cs, colsum_beta=1.0)


print('%-44s %12s %12s' % ('shape (G, M, N, K, ta, tb)', 'two launches', 'in kernel'))
for case in [(1, 4352, 600, 1800, False, False), (2, 900, 600, 4352, True, False), (3, 300, 600, 4352, True, False), (2, 600, 300, 4352, True, False),
             (2, 300, 300, 4352, True, False), (1, 300, 600, 4352, True, False), (1, 150, 300, 4352, True, False), (1, 27, 150, 4352, True, False),
             (3, 4352, 600, 300, False, False), (1, 16, 81, 4096, True, False), (2, 128, 64, 7168, True, False), (2, 192, 128, 7168, True, False),
             (1, 13056, 1800, 600, False, True)]:
    fn = dense(*case)
    lib.ha2g_splitk_in_kernel(0)
    lib.ha2g_gemm_debug_plane_ksplit(0, 0)
    a = t_us(fn)
    lib.ha2g_splitk_in_kernel(1)
    lib.ha2g_gemm_debug_plane_ksplit(1, 0)
    b = t_us(fn)
    fl = 2.0 * case[0] * case[1] * case[2] * case[3]
    print('%-44s %8.1f us %8.1f us   %6.1f -> %6.1f TFLOP/s' % (case, a, b, fl / a * 1e-6, fl / b * 1e-6))
print('plane GEMM dX [4352 x 600] x K = 1800 by forced k slices (in-kernel reduction):')
fn = dense(1, 4352, 600, 1800, False, False)
for ks in (1, 2, 3, 4, 5, 6):
    lib.ha2g_gemm_debug_plane_ksplit(1, ks)
    print('   k slices %d: %8.1f us' % (ks, t_us(fn)))
lib.ha2g_gemm_debug_plane_ksplit(1, 0)

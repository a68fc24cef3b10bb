"""GRU weight-gradient products of one layer (B*T = 4352 rows, H = 300, both directions), alone: the grouped split-K GEMMs of the step against the plane
GEMM on transposed piece planes (ops._gemm_planes): is the transposing split + the plane kernel cheaper than the in-kernel split?"""
import sys, torch
sys.path.insert(0, '.')
from ha2g_amd import ops
from ha2g_amd._lib import lib
dev = torch.device('cuda:0')
R, H, K = 4352, 300, 600
dg = torch.randn(R, 8 * H, device=dev)
x2 = torch.randn(R, K, device=dev)
hp2 = torch.randn(R, 2 * H, device=dev)


def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


out_ih = [torch.zeros(3 * H, K, device=dev) for _ in range(2)]
out_hh = [torch.zeros(3 * H, H, device=dev) for _ in range(2)]
cs_ih = [torch.zeros(3 * H, device=dev) for _ in range(2)]
cs_hh = [torch.zeros(3 * H, device=dev) for _ in range(2)]


def grouped():
    A = [[dg[:, 3 * H * d:3 * H * d + 3 * H] for d in range(2)], [dg[:, 3 * H * d:3 * H * d + 2 * H] for d in range(2)], [dg[:, 6 * H + H * d:6 * H + H * (d + 1)] for d in range(2)]]
    Bm = [[x2, x2], [hp2[:, d * H:(d + 1) * H] for d in range(2)], [hp2[:, d * H:(d + 1) * H] for d in range(2)]]
    Cm = [[out_ih[d] for d in range(2)], [out_hh[d][:2 * H] for d in range(2)], [out_hh[d][2 * H:] for d in range(2)]]
    cs = [[cs_ih[d] for d in range(2)], [cs_hh[d][:2 * H] for d in range(2)], [cs_hh[d][2 * H:] for d in range(2)]]
    for i in range(3):
        ops.gemm_grouped(A[i], Bm[i], transa=True, out=Cm[i], beta=1.0, colsum_out=cs[i], colsum_beta=1.0)


def grouped_i(i):
    A = [[dg[:, 3 * H * d:3 * H * d + 3 * H] for d in range(2)], [dg[:, 3 * H * d:3 * H * d + 2 * H] for d in range(2)], [dg[:, 6 * H + H * d:6 * H + H * (d + 1)] for d in range(2)]]
    Bm = [[x2, x2], [hp2[:, d * H:(d + 1) * H] for d in range(2)], [hp2[:, d * H:(d + 1) * H] for d in range(2)]]
    Cm = [[out_ih[d] for d in range(2)], [out_hh[d][:2 * H] for d in range(2)], [out_hh[d][2 * H:] for d in range(2)]]
    cs = [[cs_ih[d] for d in range(2)], [cs_hh[d][:2 * H] for d in range(2)], [cs_hh[d][2 * H:] for d in range(2)]]
    ops.gemm_grouped(A[i], Bm[i], transa=True, out=Cm[i], beta=1.0, colsum_out=cs[i], colsum_beta=1.0)


print('grouped split-K GEMMs, one layer (3 launches + reduces): %.1f us' % timeit(grouped))
for i, n in enumerate(('dW_ih [900x600]x2', 'dW_hh rz [600x300]x2', 'dW_hh n [300x300]x2')):
    print('   %-24s %.1f us' % (n, timeit(lambda: grouped_i(i))))

big = torch.zeros(6 * H, K, device=dev)
ops.PLANE_GEMM_MIN_FLOP = 1e8
t_all = timeit(lambda: ops.gemm(dg[:, :6 * H], x2, transa=True, out=big))
print('plane GEMM dW_ih both directions [1800x600]xK=4352 incl. the two transposing splits: %.1f us' % t_all)
pa = ops._planes_2d(dg[:, :6 * H], True)
pb = ops._planes_2d(x2, True)
print('   transposing split of dg[:, :6H]: %.1f us, of x: %.1f us' % (timeit(lambda: ops._planes_2d(dg[:, :6 * H], True)), timeit(lambda: ops._planes_2d(x2, True))))
ws = ops.workspace(dev)
print('   the product alone: %.1f us' % timeit(lambda: ops.check(lib.ha2g_gemm_planes_np_f32(pa.data_ptr(), pa.stride(0), pa.shape[2], pb.data_ptr(), pb.stride(0), pb.shape[2], 3,
                                                                                             6 * H, K, R, 0.0, big.data_ptr(), big.stride(0), 0, 0, ws.data_ptr(), ws.numel() * 4, ops._stream()))))
for f in (2, 3, 4, 5, 6, 7, 8):
    lib.ha2g_gemm_debug_plane_ksplit(0, f)
    print('      forced %d k slices: %.1f us' % (f, timeit(lambda: ops.check(lib.ha2g_gemm_planes_np_f32(pa.data_ptr(), pa.stride(0), pa.shape[2], pb.data_ptr(), pb.stride(0), pb.shape[2], 3,
                                                                                                      6 * H, K, R, 0.0, big.data_ptr(), big.stride(0), 0, 0, ws.data_ptr(), ws.numel() * 4, ops._stream())))))
lib.ha2g_gemm_debug_plane_ksplit(0, 0)
hh = torch.zeros(3 * H, H, device=dev)
ph = ops._planes_2d(hp2[:, :H], True)
pg = ops._planes_2d(dg[:, :3 * H], True)
for f in (0, 4, 6, 8, 10, 12, 16):
    lib.ha2g_gemm_debug_plane_ksplit(0, f)
    print('   plane product [900x300]xK=4352 (one direction of dW_hh), %2d k slices: %.1f us' % (f, timeit(lambda: ops.check(lib.ha2g_gemm_planes_np_f32(
        pg.data_ptr(), pg.stride(0), pg.shape[2], ph.data_ptr(), ph.stride(0), ph.shape[2], 3, 3 * H, H, R, 0.0, hh.data_ptr(), hh.stride(0), 0, 0, ws.data_ptr(), ws.numel() * 4, ops._stream())))))
lib.ha2g_gemm_debug_plane_ksplit(0, 0)
ref = dg[:, :6 * H].double().t() @ x2.double()
ops.gemm(dg[:, :6 * H], x2, transa=True, out=big)
print('plane dW_ih vs float64: %.2e' % float((big.double() - ref).abs().max() / ref.abs().max()))

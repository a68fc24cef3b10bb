"""Dense-GEMM tile / split-K sweep: the heaviest dense shapes of one train step (tools/gemm_census.py), each timed under every tile shape
(ha2g_gemm_debug_tile) x split-K count, against the heuristic's choice.  GEMM + its split-K reduce launch are timed together."""
import sys, torch
sys.path.insert(0, '.')
from ha2g_amd import ops
from ha2g_amd._lib import lib
dev = torch.device('cuda:0')
SHAPES = [  # calls, M, N, K, ta, tb
    (18, 13056, 900, 600, 0, 1), (24, 8704, 300, 600, 0, 1), (32, 4352, 600, 300, 0, 0), (32, 4352, 300, 600, 0, 1),
    (18, 900, 600, 4352, 1, 0), (32, 300, 600, 4352, 1, 0), (18, 4352, 600, 900, 0, 0), (24, 600, 300, 4352, 1, 0),
    (24, 300, 300, 4352, 1, 0), (6, 7168, 128, 192, 0, 0), (6, 3584, 128, 192, 0, 0), (2, 13056, 900, 102, 0, 1),
    (3, 8704, 150, 300, 0, 1), (6, 4352, 900, 600, 0, 1), (6, 4352, 300, 150, 0, 0), (6, 150, 300, 4352, 1, 0)]
NAMES = ['128x128', '64x128', '128x64', '64x64', '128x32', '128x192', '128x160', '64x192', '128x96']


def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


tot_h = tot_b = 0.0
for c, M, N, K, ta, tb in SHAPES:
    a = torch.randn((K, M) if ta else (M, K), device=dev); b = torch.randn((N, K) if tb else (K, N), device=dev)
    f = lambda: ops.gemm(a, b, transa=bool(ta), transb=bool(tb))
    lib.ha2g_gemm_debug_tile(-1, 0)
    ref = f().clone()
    h = timeit(f)
    res = []
    for cfg in range(9):
        for sp in ((0, 1, 2, 3, 4, 6, 8, 12, 16) if K >= 512 else (0,)):
            lib.ha2g_gemm_debug_tile(cfg, sp)
            out = f()
            err = float((out - ref).abs().max() / ref.abs().max())
            assert err < 1e-4, (cfg, sp, err)
            res.append((timeit(f), NAMES[cfg], sp))
    lib.ha2g_gemm_debug_tile(-1, 0)
    print('RAW', M, N, K, ta, tb, ' '.join('%s/%d/%.1f' % (n, sp, t) for t, n, sp in res))
    res.sort()
    tot_h += c * h; tot_b += c * res[0][0]
    fl = 2.0 * M * N * K
    print('%2d x M=%6d N=%4d K=%5d ta=%d tb=%d | heuristic %6.1f us (%5.1f TF) | best: %s' % (
        c, M, N, K, ta, tb, h, fl / h / 1e6, '  '.join('%s/s%d %.1f' % (n, s, t) for t, n, s in res[:5])))
print('per step: heuristic %.2f ms, best-of-sweep %.2f ms' % (tot_h / 1e3, tot_b / 1e3))

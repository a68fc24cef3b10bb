"""Census of the dense products of one train step: every ops.gemm / ops.gemm_grouped call of the step with its shape, its caller and its time
(HIP events on the launch stream, the split-K reduce that belongs to it included), summed per (caller, shape).  Runs bench.py's own step set-up
(same model, batch and mode) with the two wrappers patched:

    python tools/gemm_census.py [bench flags]          e.g. --expressive
"""
import json
import sys

sys.path.insert(0, '.')
import bench  # noqa: E402
from ha2g_amd import ops  # noqa: E402

_gemm, _grouped = ops.gemm, ops.gemm_grouped


def _caller():
    f = sys._getframe(2)
    names = []
    while f is not None and len(names) < 3:
        n = f.f_code.co_name
        if n not in ('<lambda>', 'launch', 'gemm', 'gemm_grouped', 'apply'):
            q = f.f_locals.get('self', None)
            names.append((type(q).__name__ + '.' if q is not None and not isinstance(q, type) else '') + n)
        f = f.f_back
    return '<'.join(names)


def gemm(a, b, transa=False, transb=False, out=None, **kw):
    M, K = (a.shape[1], a.shape[0]) if transa else a.shape
    N = b.shape[0] if transb else b.shape[1]
    key = 'G|1|%d|%d|%d|%d%d|%s' % (M, N, K, transa, transb, _caller())
    return ops.ktimer.launch(key, lambda: _gemm(a, b, transa=transa, transb=transb, out=out, **kw), 2.0 * M * N * K)


def gemm_grouped(a, b, transa=False, transb=False, out=None, **kw):
    a0 = a[0]
    b0 = b[0]
    G = len(a)
    M, K = (a0.shape[1], a0.shape[0]) if transa else a0.shape
    N = b0.shape[0] if transb else b0.shape[1]
    key = 'G|%d|%d|%d|%d|%d%d|%s' % (G, M, N, K, transa, transb, _caller())
    return ops.ktimer.launch(key, lambda: _grouped(a, b, transa=transa, transb=transb, out=out, **kw), 2.0 * G * M * N * K)


ops.gemm, ops.gemm_grouped = gemm, gemm_grouped

if __name__ == '__main__':
    import io
    import contextlib
    sys.argv = [sys.argv[0], '--steps', '6', '--warmup', '3', '--no-cpu-baseline', '--primary-only', '--launch', 'eager'] + sys.argv[1:]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    line = json.loads(buf.getvalue().strip().splitlines()[-1])
    steps = 6
    rows = []
    for k, (n, us) in line['kernel_times_us'].items():
        if not k.startswith('G|'):
            continue
        _, G, M, N, K, tt, who = k.split('|', 6)
        G, M, N, K = int(G), int(M), int(N), int(K)
        per_step = n / steps
        fl = 2.0 * G * M * N * K
        rows.append((per_step * us, per_step, us, G, M, N, K, tt, fl / us / 1e6, who))
    rows.sort(reverse=True)
    print('step %.3f ms; dense products: %d calls / step, %.2f ms / step (sum of per-call HIP-event times, both queues)' % (
        line['ms_per_step'], sum(r[1] for r in rows), sum(r[0] for r in rows) / 1e3))
    print('%8s %6s %8s %2s %6s %5s %5s %3s %7s  %s' % ('us/step', 'calls', 'us/call', 'G', 'M', 'N', 'K', 'tt', 'TFLOP/s', 'caller'))
    for r in rows:
        print('%8.1f %6.1f %8.1f %2d %6d %5d %5d %3s %7.1f  %s' % r)
    print('other timed launches (us/step, calls/step, us/call):')
    for k, (n, us) in sorted(line['kernel_times_us'].items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        if not k.startswith('G|'):
            print('%8.1f %6.1f %8.1f  %s' % (n / steps * us, n / steps, us, k))

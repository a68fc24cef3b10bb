"""Census of the dense GEMM calls of one train step (shape, transposes, count) with each shape timed in isolation."""
import sys, collections, torch
sys.path.insert(0, '.')
from bench import Vocab
from ha2g_amd import ops, procedural as proc
from ha2g_amd.config import hierarchy_args
from ha2g_amd.train import HierarchyTrainer
dev = torch.device('cuda:0')
args = hierarchy_args()
tr = HierarchyTrainer(args, Vocab(20000), Vocab(1371), 27, dev)
text, spec, target, vid = (torch.from_numpy(x).to(dev) for x in proc.make_batch(128, 27, 20000, 1371, 1234))
for _ in range(2): tr.train_iter(11, text, spec, target, vid)
cnt = collections.Counter()
orig = ops.gemm
def logged(a, b, transa=False, transb=False, **kw):
    M = a.shape[1] if transa else a.shape[0]; K = a.shape[0] if transa else a.shape[1]
    N = b.shape[0] if transb else b.shape[1]
    cnt[(M, N, K, bool(transa), bool(transb))] += 1
    return orig(a, b, transa=transa, transb=transb, **kw)
ops.gemm = logged
import ha2g_amd.wav_engine as we, ha2g_amd.hierarchy_net as hn
tr.train_iter(11, text, spec, target, vid)
ops.gemm = orig
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
rows = []
for (M, N, K, ta, tb), c in cnt.items():
    a = torch.randn((K, M) if ta else (M, K), device=dev); b = torch.randn((N, K) if tb else (K, N), device=dev)
    us = timeit(lambda: orig(a, b, transa=ta, transb=tb))
    rows.append((c * us, c, us, M, N, K, ta, tb))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print('distinct shapes %d, calls %d, isolated total %.2f ms' % (len(rows), sum(r[1] for r in rows), tot / 1e3))
for t, c, us, M, N, K, ta, tb in rows[:28]:
    print('%7.0f us total | %3d x %6.1f us | M=%6d N=%5d K=%6d ta=%d tb=%d | %5.1f TF' % (t, c, us, M, N, K, ta, tb, 2.0 * M * N * K / us / 1e6))

"""Capture one whole train step (forward, both backward stages, 6 Adam updates; ~2 600 kernels on 2 streams) into a hipGraph
and replay it: checks that the step is capturable as designed (device-resident RNG / Adam counters, no host syncs) and
measures eager vs replay time."""
import sys, time, torch
sys.path.insert(0, '.')
from bench import Vocab
from ha2g_amd import ops, procedural as proc
from ha2g_amd.config import hierarchy_args
from ha2g_amd.train import HierarchyTrainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device('cuda:0')
torch.manual_seed(0)
tr = HierarchyTrainer(hierarchy_args(), Vocab(20000), Vocab(1371), 27, dev)
text, spec, target, vid = (torch.from_numpy(x).to(dev) for x in proc.make_batch(B, 27, 20000, 1371, 1234))
for _ in range(3):
    tr.train_iter(11, text, spec, target, vid)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    r = tr.train_iter(11, text, spec, target, vid)
torch.cuda.synchronize()
print('eager: %.2f ms/step' % ((time.perf_counter() - t0) * 100), {k: round(v, 4) for k, v in r.items()})
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        tr.train_iter(11, text, spec, target, vid, return_tensors=True)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    names, packed = tr.train_iter(11, text, spec, target, vid, return_tensors=True)
torch.cuda.synchronize()
print('captured')
vals = []
for _ in range(3):
    g.replay()
    vals.append(packed.tolist())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
print('graph replay: %.2f ms/step' % ((time.perf_counter() - t0) * 100))
for v in vals:
    print(dict(zip(names, [round(x, 4) for x in v])))
print('cluster hand-off timeouts:', ops.gru_cluster_error(dev))

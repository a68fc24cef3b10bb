"""Census of colsum (bias-gradient) calls in one step."""
import sys, collections, torch
sys.path.insert(0, '.')
from bench import Vocab
from ha2g_amd import ops, procedural as proc
from ha2g_amd.config import hierarchy_args
from ha2g_amd.train import HierarchyTrainer
dev = torch.device('cuda:0')
tr = HierarchyTrainer(hierarchy_args(), Vocab(20000), Vocab(1371), 27, dev)
text, spec, target, vid = (torch.from_numpy(x).to(dev) for x in proc.make_batch(128, 27, 20000, 1371, 1234))
for _ in range(2): tr.train_iter(11, text, spec, target, vid)
cnt = collections.Counter()
orig = ops.colsum
def logged(x, out=None, beta=0.0):
    cnt[(x.shape[0], x.shape[1], x.stride(0))] += 1
    return orig(x, out=out, beta=beta)
ops.colsum = logged
import ha2g_amd.wav_engine as we
tr.train_iter(11, text, spec, target, vid)
for (r, c, ld), n in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print('%4d x rows=%6d cols=%5d ld=%5d' % (n, r, c, ld))
print('total', sum(cnt.values()))

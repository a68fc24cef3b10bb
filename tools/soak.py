"""Soak: N training steps on a fixed synthetic batch; losses must stay finite and decrease, memory must not grow."""
import sys, time, torch
sys.path.insert(0, '.')
from bench import Vocab
from ha2g_amd import ops, procedural as proc
from ha2g_amd.config import hierarchy_args
from ha2g_amd.train import HierarchyTrainer
dev = torch.device('cuda:0')
torch.manual_seed(0)
args = hierarchy_args()
tr = HierarchyTrainer(args, Vocab(20000), Vocab(1371), 27, dev)
text, spec, target, vid = (torch.from_numpy(x).to(dev) for x in proc.make_batch(128, 27, 20000, 1371, 1234))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
for i in range(n):
    epoch = 0 if i < n // 3 else 11
    r = tr.train_iter(epoch, text, spec, target, vid)
    if i % 20 == 0 or i == n - 1:
        print(i, 'epoch', epoch, {k: round(v, 4) for k, v in r.items()}, 'mem GB %.2f peak %.2f' % (torch.cuda.memory_allocated() / 1e9, torch.cuda.max_memory_allocated() / 1e9), flush=True)
    assert all(v == v and abs(v) < 1e6 for v in r.values()), r
print('cluster hand-off timeouts:', ops.gru_cluster_error(dev))

"""How long does the host need to ENQUEUE one train step vs how long the GPU needs to execute it?"""
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
for _ in range(3):
    tr.train_iter(11, text, spec, target, vid)
torch.cuda.synchronize()
for trial in range(3):
    t0 = time.perf_counter()
    names, packed = tr.train_iter(11, text, spec, target, vid, return_tensors=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('enqueue %.1f ms, then GPU still busy for %.1f ms (total %.1f ms)' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3))

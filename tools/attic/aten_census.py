"""torch-side launches of one train step (copies, cats, adds, fills ... -- everything that is not a libha2g_hip.so kernel) by op and shape."""
import sys, collections, torch
sys.path.insert(0, '.')
from bench import Vocab
from ha2g_amd import procedural as proc
from ha2g_amd.config import hierarchy_args
from ha2g_amd.train import HierarchyTrainer
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
tr = HierarchyTrainer(hierarchy_args(), Vocab(20000), Vocab(1371), 27, dev)
text, spec, target, vid = (torch.from_numpy(x).to(dev) for x in proc.make_batch(128, 27, 20000, 1371, 1234))
for _ in range(2): tr.train_iter(11, text, spec, target, vid)
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=False) as prof:
    tr.train_iter(11, text, spec, target, vid)
cnt = collections.Counter()
LAUNCHING = ('aten::copy_', 'aten::cat', 'aten::add', 'aten::add_', 'aten::fill_', 'aten::zero_', 'aten::index_select', 'aten::repeat', 'aten::stack',
             'aten::mul', 'aten::sum', 'aten::clone', 'aten::contiguous', 'aten::_to_copy', 'aten::index', 'aten::index_put_', 'aten::neg', 'aten::abs',
             'aten::mean', 'aten::masked_fill_', 'aten::where', 'aten::arange', 'aten::randperm', 'aten::zeros', 'aten::ones', 'aten::full')
for e in prof.events():
    if e.name in LAUNCHING:
        cnt[(e.name, str(e.input_shapes)[:90])] += 1
tot = collections.Counter()
for (n, s), c in cnt.items(): tot[n] += c
print(dict(tot))
for (n, s), c in cnt.most_common(45): print('%4d x %-22s %s' % (c, n, s))

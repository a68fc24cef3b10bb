"""Which Python lines of the train step issue device-to-device copies / fills / cats / adds through torch (the ~400 'torch plumbing' launches of a step)?
torch.profiler with stacks over two eager steps; grouped by the innermost ha2g_amd frame."""
import sys, collections
sys.path.insert(0, '.')
import torch
from torch.profiler import profile, ProfilerActivity
from bench import Vocab
from ha2g_amd import procedural as proc
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
torch.autograd.set_multithreading_enabled(False)
N = 2
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    for _ in range(N):
        tr.train_iter(11, text, spec, target, vid)
    torch.cuda.synchronize()
agg = collections.Counter()
shapes = {}
WATCH = ('aten::copy_', 'aten::cat', 'aten::fill_', 'aten::zero_', 'aten::add', 'aten::add_', 'aten::clone', 'aten::contiguous', 'aten::index_select', 'aten::repeat', 'aten::stack',
         'aten::mul', 'aten::zeros', 'aten::zeros_like', 'aten::sum', 'aten::_to_copy')
for ev in prof.events():
    if ev.name not in WATCH:
        continue
    frame = None
    for fr in ev.stack:
        if 'ha2g_amd' in fr or 'bench.py' in fr:
            frame = fr.strip()
            break
    if frame is None:
        frame = '(autograd engine / no repo frame)'
    key = (ev.name, frame.split('/root/repo/')[-1] if '/root/repo/' in frame else frame[-90:])
    agg[key] += 1
    shapes.setdefault(key, str(ev.input_shapes)[:70])
tot = collections.Counter()
for (name, fr), c in agg.items():
    tot[name] += c
print('per step:', {k: v / N for k, v in tot.most_common()})
for (name, fr), c in agg.most_common(70):
    print('%6.1f  %-18s %-70s %s' % (c / N, name, fr[:70], shapes[(name, fr)]))

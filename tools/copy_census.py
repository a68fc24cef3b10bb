"""Which Python lines of the train step issue torch's own small kernels (copies / fills / cats / adds: the 'torch plumbing' launches of a step)?
The calls are counted per calling line of ha2g_amd by wrapping the torch entry points for two eager steps (autograd's own accumulations -- gradient
fan-in sums -- have no Python caller and are reported by their count only)."""
import sys, collections
sys.path.insert(0, '.')
import torch
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
agg = collections.Counter()


def caller():
    f = sys._getframe(2)
    while f is not None:
        fn = f.f_code.co_filename
        if 'ha2g_amd' in fn and 'copy_census' not in fn:
            return '%s:%d %s' % (fn.split('ha2g_amd/')[-1], f.f_lineno, f.f_code.co_name)
        f = f.f_back
    return '(no ha2g_amd frame)'


def wrap(owner, name):
    orig = getattr(owner, name)

    def w(*a, **k):
        r = orig(*a, **k)
        t = r if torch.is_tensor(r) else (a[0] if a and torch.is_tensor(a[0]) else None)
        if t is None or t.is_cuda:
            agg[(name, caller(), tuple(t.shape) if t is not None else None)] += 1
        return r
    setattr(owner, name, w)
    return orig


saved = []
for owner, names in ((torch.Tensor, ('copy_', 'zero_', 'fill_', 'clone', 'contiguous', 'repeat', 'index_select', 'add_', 'add', 'mul', 'sum', '__add__', '__mul__', '__iadd__', 'to', 'float')),
                     (torch, ('cat', 'stack', 'zeros', 'zeros_like', 'ones', 'full', 'empty_like', 'randn', 'randperm', 'arange'))):
    for n in names:
        saved.append((owner, n, wrap(owner, n)))
N = 2
for _ in range(N):
    tr.train_iter(11, text, spec, target, vid)
torch.cuda.synchronize()
for owner, n, o in saved:
    setattr(owner, n, o)
skip = ('empty_like', 'contiguous', 'to', 'float')         # (no kernel when nothing changes; listed at the end)
rows = sorted(agg.items(), key=lambda kv: -kv[1])
print('calls per step with a device tensor, by calling line (contiguous / to / float / empty_like launch nothing when the tensor already qualifies):')
for (name, fr, shp), c in rows:
    if name not in skip:
        print('%6.1f  %-12s %-58s %s' % (c / N, name, fr[:58], shp))
print('--')
for (name, fr, shp), c in rows:
    if name in skip and c / N >= 4:
        print('%6.1f  %-12s %-58s %s' % (c / N, name, fr[:58], shp))

"""Census of the small pointwise launches of one train step: ops.eltwise calls by (op, numel) and the torch-side copies / cats / fills."""
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
orig = ops.eltwise
NAMES = {v: k for k, v in vars(ops).items() if k.startswith('OP_') and isinstance(v, int)}
def logged(op, a, *rest, **kw):
    import traceback
    fr = traceback.extract_stack(limit=4)[0]
    cnt[(NAMES.get(op, op), a.numel(), '%s:%d' % (fr.filename.split('/')[-1], fr.lineno))] += 1
    return orig(op, a, *rest, **kw)
ops.eltwise = logged
import ha2g_amd.wav_engine as we, ha2g_amd.hierarchy_net as hn, ha2g_amd.train_hierarchy as th
for m in (we, hn, th):
    if hasattr(m, 'eltwise'): m.eltwise = logged
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU]) as prof:
    tr.train_iter(11, text, spec, target, vid)
ops.eltwise = orig
print('eltwise calls', sum(cnt.values()))
for k, c in cnt.most_common(40): print('%4d x %s' % (c, k))
print(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=45, max_name_column_width=50))

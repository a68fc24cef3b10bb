"""Event-time the big sections of one train step (forward pieces via module hooks, backward as a whole)."""
import sys, time, torch
sys.path.insert(0, '.')
from bench import Vocab
from ha2g_amd import ops, procedural as proc, train_hierarchy as th
from ha2g_amd.config import hierarchy_args
from ha2g_amd.train import HierarchyTrainer
dev = torch.device('cuda:0')
torch.manual_seed(0)
args = hierarchy_args()
tr = HierarchyTrainer(args, Vocab(20000), Vocab(1371), 27, dev)
text, spec, target, vid = (torch.from_numpy(x).to(dev) for x in proc.make_batch(128, 27, 20000, 1371, 1234))
for _ in range(3):
    tr.train_iter(11, text, spec, target, vid)
recs = []
def ev(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); recs.append((name, e))
def wrap(mod, name):
    f = mod.forward
    def g(*a, **k):
        ev(name + ':begin'); r = f(*a, **k); ev(name + ':end'); return r
    mod.forward = g
wrap(tr.audio_encoder, 'audio_fwd'); wrap(tr.text_encoder, 'text_fwd'); wrap(tr.discriminator, 'D_fwd')
for i, g in enumerate(tr.gens): wrap(g, 'g%d_fwd' % (i + 1))
import torch.autograd
orig_bwd = torch.Tensor.backward
def bwd(self, *a, **k):
    ev('backward:begin'); r = orig_bwd(self, *a, **k); ev('backward:end'); return r
torch.Tensor.backward = bwd
for o in tr.gen_opts + [tr.audio_opt, tr.text_opt, tr.dis_opt]:
    f = o.step
    def s(f=f):
        ev('adam:begin'); f(); ev('adam:end')
    o.step = s
ev('step:begin'); tr.train_iter(11, text, spec, target, vid); ev('step:end')
torch.cuda.synchronize()
t0 = recs[0][1]
opened = {}
tot = {}
for name, e in recs:
    base, kind = name.split(':')
    if kind == 'begin': opened[base] = e
    else: tot[base] = tot.get(base, 0.0) + opened[base].elapsed_time(e)
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]): print('%-12s %7.2f ms' % (k, v))

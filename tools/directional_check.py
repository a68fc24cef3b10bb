import sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from ha2g_amd import ops, procedural as proc
from ha2g_amd.config import hierarchy_args
from ha2g_testing import SpeakerVocab
from ha2g_amd.train import HierarchyTrainer
DEV = 'cuda:0'; dev = torch.device(DEV)
class Lang: n_words, word_embedding_weights = 300, None
batch = [torch.from_numpy(x).to(DEV) for x in proc.make_batch(16, 27, 300, 20, 5)]
PN = [0.0] * 5
def run(direction=None, eps=0.0, epoch=11, drop=None):
    torch.manual_seed(3); ops.rng.seed(dev, 99)
    args = hierarchy_args(); args.learning_rate = 0.0
    if drop is not None: args.dropout_prob = drop
    tr = HierarchyTrainer(args, Lang(), SpeakerVocab(20), 27, dev)
    if drop == 0.0:
        for m in [tr.text_encoder] + [g.text_encoder for g in tr.gens]:
            m.drop.p = 0.0
            for b in m.tcn.network: b.p = 0.0
    opts = list(tr.gen_opts) + [tr.audio_opt, tr.text_opt]
    if direction is None:
        PN[:] = [float(o.flat_p.double().norm()) for o in opts]
    if direction is not None:
        for o, d in zip(opts, direction): o.flat_p.add_(d, alpha=eps)
    r = tr.train_iter(epoch, *batch); torch.cuda.synchronize()
    return r, [o.flat_g.clone() for o in opts]
import os
DROP = None if os.environ.get('DD_DROP', '1') != '0' else 0.0
names = ['g1', 'g2', 'g3', 'audio', 'text']
for epoch in (11, 0):
    r0, g = run(epoch=epoch, drop=DROP)
    keys = [k for k in r0 if k != 'dis']
    for i, nm in enumerate(names):
        gn = float(g[i].double().norm())
        d = [torch.zeros_like(t) for t in g]; d[i] = g[i] / gn
        for rel in (1e-3, 2.5e-4):                                 # step = rel x the norm of the module's parameters
            eps = rel * PN[i]
            rp, _ = run(d, eps, epoch, DROP); rm, _ = run(d, -eps, epoch, DROP)
            parts = {k: (rp[k] - rm[k]) / (2 * eps) for k in keys}
            fd = sum(parts.values())
            print('epoch %2d %-5s step %.1e |theta| = %.3g: |g| %.5f fd %.5f ratio %.4f  ' % (epoch, nm, rel, eps, gn, fd, fd / gn) + ' '.join('%s %.4f' % (k, v) for k, v in parts.items() if abs(v) > 5e-5))

"""Full train_iter_hierarchy on the GPU vs the reference-generated step fixtures: two consecutive steps
(epoch 0 = warm-up phase, epoch 11 = GAN phase) -- loss dict, accumulated gradients, Adam-updated parameters,
BatchNorm running statistics -- for the fused-chain schedule and the literal three-pass schedule."""
import pytest
import torch

from ha2g_amd import procedural as proc
from ha2g_amd import train_hierarchy as th
from ha2g_amd.config import CASES
from ha2g_amd.optim import FusedAdam
from ha2g_amd.testing import Checker, EpsInjector, batch_for, build_modules, named_state

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('fuse', [True, False])
@pytest.mark.parametrize('name', ['small', 'cfg1'])
def test_train_step(golden, name, fuse):
    case, g = CASES[name], golden(name)
    ck = Checker(g)
    args, gens, dis, aud, txt = build_modules(case, DEV)
    text, spec, target, vid = (t.to(DEV) for t in batch_for(case))
    lr = float(args.learning_rate)
    g_opts = [FusedAdam(m.parameters(), lr=lr) for m in gens]
    dis_opt = FusedAdam(dis.parameters(), lr=lr * args.discriminator_lr_weight)
    aud_opt, txt_opt = FusedAdam(aud.parameters(), lr=lr), FusedAdam(txt.parameters(), lr=lr)
    EpsInjector(gens, case['seed'], case['B'])
    perm = torch.from_numpy(proc.fixed_perm(case['B'], case['seed'])).to(DEV)
    mods = dict(g1=gens[0], g2=gens[1], g3=gens[2], dis=dis, audio=aud, text=txt)
    old = th.FUSE_CHAINS, th.randperm_source
    th.FUSE_CHAINS, th.randperm_source = fuse, (lambda n, device: perm)
    try:
        for si, epoch in enumerate((0, 11)):
            ret = th.train_iter_hierarchy(args, epoch, text, spec, target, vid, *gens, dis, aud, txt, *g_opts, dis_opt,
                                          aud_opt, txt_opt)
            sd, grads = named_state(mods)
            if epoch == 0:                      # the reference's D has .grad None during warm-up (never backpropagated)
                grads = {k: v for k, v in grads.items() if not k.startswith('dis.')}
            ck.step(si, ret, grads, sd)
    finally:
        th.FUSE_CHAINS, th.randperm_source = old


@pytest.mark.parametrize('fuse', [True, False])
def test_train_step_expressive(golden, fuse):
    """6-level TED-Expressive twin (train_hierarchy_expressive.py:124-483): P=126, off-by-one head scatter, palm normals,
    eps-free contrastive, 9 modules / 9 optimizers."""
    from ha2g_amd import schema
    case, g = CASES['expr_small'], golden('expr_small')
    ck = Checker(g)
    dims = schema.EXPRESSIVE_POSE_DIMS
    args, gens, dis, aud, txt = build_modules(case, DEV, dims)
    text, spec, target, vid = (t.to(DEV) for t in batch_for(case, P=126))
    lr = float(args.learning_rate)
    g_opts = [FusedAdam(m.parameters(), lr=lr) for m in gens]
    dis_opt = FusedAdam(dis.parameters(), lr=lr * args.discriminator_lr_weight)
    aud_opt, txt_opt = FusedAdam(aud.parameters(), lr=lr), FusedAdam(txt.parameters(), lr=lr)
    EpsInjector(gens, case['seed'], case['B'])
    perm = torch.from_numpy(proc.fixed_perm(case['B'], case['seed'])).to(DEV)
    mods = {'g%d' % (i + 1): m for i, m in enumerate(gens)}
    mods.update(dis=dis, audio=aud, text=txt)
    old = th.FUSE_CHAINS, th.randperm_source
    th.FUSE_CHAINS, th.randperm_source = fuse, (lambda n, device: perm)
    try:
        for si, epoch in enumerate((0, 11)):
            ret = th.train_iter_hierarchy_expressive(args, epoch, text, spec, target, vid, *gens, dis, aud, txt, *g_opts, dis_opt,
                                                     aud_opt, txt_opt)
            sd, grads = named_state(mods)
            if epoch == 0:
                grads = {k: v for k, v in grads.items() if not k.startswith('dis.')}
            ck.step(si, ret, grads, sd)
    finally:
        th.FUSE_CHAINS, th.randperm_source = old


def test_training_loop_reduces_loss_and_is_deterministic():
    """A few real optimisation steps (dropout on, default init, HierarchyTrainer = the reference's train_epochs set-up):
    losses stay finite, the regression loss falls on a fixed batch, memory does not grow, no GRU hand-off times out,
    and two identically seeded runs are bitwise equal (no float atomics anywhere on the path)."""
    from ha2g_amd import ops
    from ha2g_amd.config import hierarchy_args
    from ha2g_amd.testing import SpeakerVocab
    from ha2g_amd.train import HierarchyTrainer

    class Lang:
        n_words, word_embedding_weights = 300, None

    def run():
        torch.manual_seed(3)
        ops.rng.seed(torch.device(DEV), 99)
        args = hierarchy_args()
        tr = HierarchyTrainer(args, Lang(), SpeakerVocab(20), 27, torch.device(DEV))
        text, spec, target, vid = (torch.from_numpy(x).to(DEV) for x in proc.make_batch(16, 27, 300, 20, 5))
        hist, mem = [], []
        for i in range(14):
            r = tr.train_iter(0 if i < 4 else 11, text, spec, target, vid)
            assert all(v == v and abs(v) < 1e6 for v in r.values()), r
            hist.append(r)
            mem.append(torch.cuda.memory_allocated())
        return hist, mem, tr.gens[2].out[2].weight.detach().clone()

    h1, m1, w1 = run()
    assert h1[-1]['loss'] < 0.95 * h1[0]['loss'], (h1[0], h1[-1])
    assert set(h1[0]) == {'loss', 'KLD', 'DIV_REG', 'c_pos', 'c_neg', 'phy'} and 'gen' in h1[-1] and 'dis' in h1[-1]
    assert abs(h1[-1]['dis'] - 1.386) < 0.1                       # 2 ln 2 early in GAN training (reference log line 209)
    assert m1[-1] <= m1[6] * 1.05 + (1 << 20)
    assert ops.gru_cluster_error(torch.device(DEV)) == 0
    h2, _, w2 = run()
    assert h1 == h2 and torch.equal(w1, w2)


def test_whole_step_hipgraph_capture_matches_eager():
    """The whole train step (forward, D phase, both backward stages, side-stream weight gradients, 6 Adam updates; ~2 000
    kernels on 2 streams) captures into ONE hipGraph: dropout / Adam counters live in device memory, nothing on the path
    syncs with the host.  With the random draws pinned (constant eps, fixed permutation, dropout off) three replays equal
    three eager steps of an identically initialised trainer bit for bit."""
    from ha2g_amd import ops
    from ha2g_amd.config import hierarchy_args
    from ha2g_amd.testing import SpeakerVocab, no_dropout
    from ha2g_amd.train import HierarchyTrainer
    dev = torch.device(DEV)

    class Lang:
        n_words, word_embedding_weights = 120, None

    B = 4
    text, spec, target, vid = (torch.from_numpy(x).to(DEV) for x in proc.make_batch(B, 27, 120, 9, 7))
    eps_const = torch.from_numpy(proc.tensor_for('in.eps', (3 * B, 16), 11)).to(DEV)
    perm = torch.from_numpy(proc.fixed_perm(B, 11)).to(DEV)

    def make():
        torch.manual_seed(5)
        ops.rng.seed(dev, 77)
        tr = HierarchyTrainer(hierarchy_args(hidden_size=32, n_layers=2), Lang(), SpeakerVocab(9), 27, dev)
        for m in tr.modules():
            no_dropout(m)
        for g in tr.gens:
            g.eps_source = lambda shape, device: eps_const[:shape[0]]
        return tr

    old = th.randperm_source
    th.randperm_source = lambda n, device: perm
    try:
        tr_e = make()
        eager = []
        for _ in range(5):
            names, packed = tr_e.train_iter(11, text, spec, target, vid, return_tensors=True)
            eager.append(packed.clone())
        tr_g = make()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                                   # warm-up off the default stream (allocations, workspaces)
            for _ in range(2):
                tr_g.train_iter(11, text, spec, target, vid, return_tensors=True)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            gnames, gpacked = tr_g.train_iter(11, text, spec, target, vid, return_tensors=True)
        replays = []
        for _ in range(2):                                           # the capture itself does not execute: steps 3 and 4 are replays
            graph.replay()
            replays.append(gpacked.clone())
        torch.cuda.synchronize()
    finally:
        th.randperm_source = old
    assert gnames == names
    for got, ref in zip(replays, eager[2:4]):
        assert torch.equal(got, ref), (got.tolist(), ref.tolist())
    assert ops.gru_cluster_error(dev) == 0

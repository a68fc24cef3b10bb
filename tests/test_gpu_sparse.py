"""Sparse embedding gradients + lazy row-wise Adam (csrc/sparse.hip) are BIT-IDENTICAL to the dense path: dense
torch.optim.Adam semantics move every row of the table on every step (m <- b1 m, v <- b2 v, p -= ...), also rows without gradient;
the row-wise optimizer replays exactly those updates when a row is next read or updated."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _toy(sparse, V=50, C=300):
    from ha2g_amd import hierarchy_net as hn
    from ha2g_amd.optim import FusedAdam
    torch.manual_seed(3)
    emb, lin = hn.Embedding(V, C).to(DEV), hn.Linear(C, 8).to(DEV)
    opt = FusedAdam(list(emb.parameters()) + list(lin.parameters()), lr=5e-4, betas=(0.5, 0.999), sparse=[emb.weight] if sparse else ())
    return emb, lin, opt


def test_unique_tokens_compaction():
    from ha2g_amd._lib import check, lib
    from ha2g_amd.ops import _stream
    r = np.random.Generator(np.random.PCG64(1))
    for n, V in ((136, 60), (4352, 20000), (5, 10), (8704, 300)):
        tok = torch.from_numpy(np.where(r.random(n) < 0.7, 0, r.integers(1, V, n))).to(DEV)
        mp = torch.full((V,), 2 ** 31 - 1, dtype=torch.int32, device=DEV)
        uniq, remap = torch.empty(n + 1, dtype=torch.int64, device=DEV), torch.empty(n, dtype=torch.int64, device=DEV)
        cpos, count = torch.empty(n, dtype=torch.int32, device=DEV), torch.empty(1, dtype=torch.int32, device=DEV)
        check(lib.ha2g_unique_tokens(tok.data_ptr(), n, mp.data_ptr(), cpos.data_ptr(), uniq.data_ptr(), remap.data_ptr(), count.data_ptr(), _stream()))
        c = int(count.item())
        t = tok.cpu().numpy()
        first = [0] + [int(x) for x in dict.fromkeys(int(v) for v in t if v != 0)]            # slot 0 = padding id, then first-occurrence order
        assert c == len(first) and uniq[:c].cpu().tolist() == first
        assert torch.equal(uniq[:c][remap], tok)
        assert int((mp != 2 ** 31 - 1).sum()) == 0                                             # scratch restored


def test_sparse_adam_is_bitwise_dense_adam():
    V, B, T = 50, 6, 34
    r = np.random.Generator(np.random.PCG64(5))
    nets = {k: _toy(k) for k in (False, True)}
    # token batches with long gaps: ids 1..9 appear in steps 0-1, disappear for 5 steps, return in step 7; id 30 only in step 3; most ids never
    batches = []
    for step in range(10):
        t = np.zeros((B, T), np.int64)
        pool = list(range(1, 10)) if step in (0, 1, 7, 8) else ([30] if step == 3 else list(range(12, 18)))
        for b in range(B):
            pos = r.choice(T, size=6, replace=False)
            t[b, pos] = r.choice(pool, size=6)
        batches.append(torch.from_numpy(t).to(DEV))
    wy = torch.from_numpy(r.standard_normal((B, T, 8)).astype(np.float32)).to(DEV)
    for step, tok in enumerate(batches):
        read = {}
        for k, (emb, lin, opt) in nets.items():
            opt.zero_grad()
            e = emb(tok)
            read[k] = e.detach().clone()
            (lin(e) * wy).sum().backward()
            opt.step()
        assert torch.equal(read[False], read[True]), 'rows read at step %d differ' % step       # lazily caught-up rows == dense rows
    d_emb, _, d_opt = nets[False]
    s_emb, _, s_opt = nets[True]
    touched = torch.unique(torch.cat([b.reshape(-1) for b in batches]))
    assert not torch.equal(d_emb.weight.detach(), s_emb.weight.detach())                       # stale rows exist before the sync ...
    s_opt.sync_sparse()
    assert torch.equal(d_emb.weight.detach(), s_emb.weight.detach())                           # ... and none after it
    tb = s_opt.sparse_tables[0]
    off = 0                                                                                    # dense moments of the table = first V*C entries
    assert torch.equal(d_opt.flat_m[off:off + V * 300].view(V, 300), tb.m) and torch.equal(d_opt.flat_v[off:off + V * 300].view(V, 300), tb.v)
    assert int((tb.last[touched] == 10).sum()) == touched.numel()
    assert torch.equal(d_opt.flat_p[V * 300:], s_opt.flat_p)                                   # the dense remainder is the same flat buffer content


def test_merged_rank_lists_equal_dense_mean():
    """Two 'ranks' worth of compact gradients merged (ops.merge_rows, as after ddp.gather_sparse_rows) == mean of the dense gradients."""
    from ha2g_amd import ops
    V, C = 40, 300
    r = np.random.Generator(np.random.PCG64(7))
    dense = torch.zeros(V, C, dtype=torch.float64)
    ids_l, rows_l = [], []
    for rank in range(2):
        ids = np.concatenate([[0], r.choice(np.arange(1, V), size=9, replace=False), [0, 0]])   # padded tail: id 0, zero rows
        rows = r.standard_normal((12, C)).astype(np.float32)
        rows[10:] = 0
        dense.index_add_(0, torch.from_numpy(ids), torch.from_numpy(rows).double() * 0.5)
        ids_l.append(torch.from_numpy(ids)); rows_l.append(torch.from_numpy(rows) * 0.5)
    mp = torch.full((V,), 2 ** 31 - 1, dtype=torch.int32, device=DEV)
    uniq, count, vals = ops.merge_rows(torch.cat(ids_l).to(DEV), torch.cat(rows_l).to(DEV), mp)
    c = int(count.item())
    got = torch.zeros(V, C, dtype=torch.float64)
    got[uniq[:c].cpu()] = vals[:c].double().cpu()
    assert float((got - dense).abs().max()) < 1e-6
    assert len(set(uniq[:c].cpu().tolist())) == c


def test_train_step_with_sparse_tables_equals_dense_step():
    """Four whole train steps (warm-up, GAN, GAN, GAN) of two identically initialised trainers, dense vs row-wise word-embedding tables:
    identical loss dicts and, after sync_sparse(), bit-identical parameters."""
    from ha2g_amd import ops, procedural as proc, train_hierarchy as th
    from ha2g_amd.config import hierarchy_args
    from ha2g_testing import SpeakerVocab, no_dropout
    from ha2g_amd.train import HierarchyTrainer
    dev = torch.device(DEV)

    class Lang:
        n_words, word_embedding_weights = 120, None

    B = 4
    eps_const = torch.from_numpy(proc.tensor_for('in.eps', (3 * B, 16), 11)).to(DEV)
    perm = torch.from_numpy(proc.fixed_perm(B, 11)).to(DEV)

    def run(sparse):
        torch.manual_seed(5)
        ops.rng.seed(dev, 77)
        tr = HierarchyTrainer(hierarchy_args(hidden_size=32, n_layers=2), Lang(), SpeakerVocab(9), 27, dev, sparse_embeddings=sparse)
        for m in tr.modules():
            no_dropout(m)
        for g in tr.gens:
            g.eps_source = lambda shape, device: eps_const[:shape[0]]
        rets = []
        for i, ep in enumerate((0, 11, 11, 11)):
            text, spec, target, vid = (torch.from_numpy(x).to(DEV) for x in proc.make_batch(B, 27, 120, 9, 7 + i))
            rets.append(tr.train_iter(ep, text, spec, target, vid))
        tr.sync_sparse()
        params = torch.cat([p.detach().reshape(-1) for m in tr.modules() for p in m.parameters()])
        return rets, params

    old = th.randperm_source
    th.randperm_source = lambda n, device: perm
    try:
        r_d, p_d = run(False)
        r_s, p_s = run(True)
    finally:
        th.randperm_source = old
    assert r_d == r_s, (r_d, r_s)
    assert torch.equal(p_d, p_s)

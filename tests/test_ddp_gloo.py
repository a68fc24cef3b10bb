"""world_size-2 data-parallel test on CPU (gloo): the gradient-averaging path the GPUs use (ha2g_amd.ddp) applied to
per-rank shard gradients computed by the CPU oracle must equal the mean of the per-shard gradients computed in one
process, and replicas that start identical stay identical after the update."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_grads(rank, world):
    """Oracle gradients of the text encoder + discriminator losses on this rank's shard (cheap, deterministic)."""
    from ha2g_amd import procedural as proc, schema
    from ha2g_amd.config import CASES
    from ha2g_amd.ddp import rank_seed
    from oracle import ha2g_oracle as O
    case = dict(CASES['small'], B=2)
    sch = {}
    sch.update(schema.text_encoder_schema(case['n_words'], case['hidden_size'], case['n_layers'], p='text.'))
    sch.update(schema.discriminator_schema(27, 'dis.'))
    sd = schema.procedural_state(sch, case['seed'])
    text, _, target, _ = map(torch.from_numpy, proc.make_batch(case['B'], 27, case['n_words'], case['n_spk'], rank_seed(100, rank)))
    ps = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and not k.endswith(('running_mean', 'running_var'))
          and '.net.' not in k}
    y = O.text_encoder_tcn(text, sd, 'text.', case['n_layers'])
    d = O.conv_discriminator(target, sd, 'dis.')
    loss = (y ** 2).mean() - torch.log(d + 1e-8).mean()
    gs = torch.autograd.grad(loss, list(ps.values()))
    return ps, dict(zip(ps.keys(), gs))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ha2g_amd import ddp
    torch.set_num_threads(2)
    ps, grads = _shard_grads(rank, world)
    flat = torch.cat([g.reshape(-1) for g in grads.values()])
    # replicas start from rank 0's parameters
    pflat = torch.cat([p.detach().reshape(-1) for p in ps.values()]) + (0.0 if rank == 0 else 1.0)
    ddp.broadcast_([pflat], 0)
    assert ddp.active()
    ddp.average_(flat)
    opt_like = pflat - 0.1 * flat
    gathered = [torch.empty_like(opt_like) for _ in range(world)]
    dist.all_gather(gathered, opt_like)
    if rank == 0:
        torch.save({'avg': flat, 'replicas_equal': all(torch.equal(gathered[0], g) for g in gathered)}, out)
    dist.destroy_process_group()


def test_two_rank_gradient_average(tmp_path):
    out = str(tmp_path / 'r0.pt')
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    assert res['replicas_equal']
    ref = []
    for r in range(2):
        _, g = _shard_grads(r, 2)
        ref.append(torch.cat([x.reshape(-1) for x in g.values()]))
    expect = (ref[0] + ref[1]) / 2
    assert not torch.equal(ref[0], ref[1])                       # shards really differ
    err = float((res['avg'] - expect).abs().max() / expect.abs().max())
    assert err < 1e-6, err


def test_single_process_is_inactive():
    from ha2g_amd import ddp
    assert not ddp.active()
    assert ddp.rank_seed(10, 3) == 13


def _flat_worker(rank, world, port, out):
    """Real FusedAdam flat-buffer layout on CPU tensors (only .step() needs the GPU): gradients land in the flat views through
    autograd, ddp.average_module_grads_ / FusedAdam.allreduce_grads average them with ONE collective per optimizer."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ha2g_amd import ddp
    from ha2g_amd.optim import FusedAdam
    torch.manual_seed(3)                                          # identical replicas
    net_a = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    net_b = torch.nn.Conv2d(2, 3, 3).to(memory_format=torch.channels_last)
    frozen = torch.nn.Parameter(torch.full((4,), float(rank)), requires_grad=False)
    opts = [FusedAdam(net_a.parameters()), FusedAdam(list(net_b.parameters()))]
    assert all(p.grad.data_ptr() >= o.flat_g.data_ptr() for o in opts for p in o.param_groups[0]['params'])
    g = torch.Generator().manual_seed(100 + rank)                 # per-rank shard
    x, img = torch.randn(6, 7, generator=g), torch.randn(2, 2, 5, 5, generator=g)
    for o in opts:
        o.zero_grad()
    (net_a(x).pow(2).mean() + net_b(img).pow(2).mean()).backward()
    local = [o.flat_g.clone() for o in opts]
    calls = []
    orig = dist.all_reduce
    dist.all_reduce = lambda t, *a, **k: (calls.append(t.numel()), orig(t, *a, **k))[1]
    ddp.average_module_grads_(opts[:1])
    w = opts[1].allreduce_grads(async_op=True)
    w.wait()
    dist.all_reduce = orig
    assert calls == [opts[0].total, opts[1].total]                # one collective per module, on the whole flat buffer
    # the parameter's .grad views see the averaged values (they alias the flat buffer)
    assert torch.equal(net_a[0].weight.grad.reshape(-1), opts[0].flat_g[:35])
    ddp.broadcast_([frozen.data], 0)
    gathered = [[torch.empty_like(l) for _ in range(world)] for l in local]
    for l, gl in zip(local, gathered):
        dist.all_gather(gl, l)
    if rank == 0:
        torch.save({'avg': [o.flat_g for o in opts], 'locals': gathered, 'frozen': frozen.data}, out)
    else:
        assert float(frozen.sum()) == 0.0                          # adopted rank 0's frozen parameter
    dist.destroy_process_group()


def test_flat_buffer_average_two_ranks(tmp_path):
    out = str(tmp_path / 'flat.pt')
    mp.spawn(_flat_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out)
    for avg, (l0, l1) in zip(res['avg'], res['locals']):
        assert not torch.equal(l0, l1)
        assert torch.allclose(avg, (l0 + l1) / 2, rtol=1e-6, atol=1e-9)


def _sparse_worker(rank, world, port, out):
    """ddp.gather_sparse_rows on CPU tensors: only max-count rows travel, invalid tail entries come back as (id 0, zero row), rows pre-scaled."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ha2g_amd import ddp
    g = torch.Generator().manual_seed(50 + rank)
    cap, C = 16, 12
    n = 5 if rank == 0 else 9                                   # ranks hold different numbers of distinct rows
    ids = torch.zeros(cap, dtype=torch.int64)
    ids[1:n] = torch.randperm(30, generator=g)[:n - 1] + 1      # slot 0 = padding id 0
    rows = torch.randn(cap, C, generator=g)
    rows[n:] = 7.0                                              # garbage past the count must not travel
    ids_all, rows_all = ddp.gather_sparse_rows(ids, torch.tensor([n], dtype=torch.int32), rows)
    assert ids_all.shape == (2 * 9,) and rows_all.shape == (2 * 9, C)
    if rank == 0:
        torch.save({'ids_all': ids_all, 'rows_all': rows_all, 'ids0': ids, 'rows0': rows}, out)
    dist.destroy_process_group()


def test_sparse_row_gather_two_ranks(tmp_path):
    out = str(tmp_path / 'sp.pt')
    mp.spawn(_sparse_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out)
    ids_all, rows_all = res['ids_all'], res['rows_all']
    assert torch.equal(ids_all[:5], res['ids0'][:5]) and int(ids_all[5:9].abs().sum()) == 0      # rank 0's block: 5 valid, 4 padded
    assert torch.allclose(rows_all[:5], res['rows0'][:5] * 0.5) and float(rows_all[5:9].abs().sum()) == 0.0
    assert int((ids_all[9:] != 0).sum()) == 8                                                     # rank 1: 9 valid (slot 0 + 8 ids)


def _bn_worker(rank, world, port, out):
    """ddp.sync_bn_stats_: BatchNorm running statistics diverge per rank during data-parallel training (each rank normalises its own shard);
    'mean' averages them with ONE collective, 'rank0' adopts rank 0's; num_batches_tracked and the affine parameters are not touched."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ha2g_amd import ddp
    torch.manual_seed(1)
    mods = [torch.nn.Sequential(torch.nn.Conv1d(3, 4, 3), torch.nn.BatchNorm1d(4)), torch.nn.BatchNorm2d(5)]
    g = torch.Generator().manual_seed(10 + rank)
    mods[0].train(); mods[1].train()
    mods[0](torch.randn(6, 3, 9, generator=g) * (1 + rank))      # per-rank shards -> per-rank running statistics
    mods[1](torch.randn(2, 5, 4, 4, generator=g) + rank)
    local = [b.clone() for b in ddp.bn_buffers(mods)]
    assert len(local) == 4
    calls = []
    orig = dist.all_reduce
    dist.all_reduce = lambda t, *a, **k: (calls.append(t.numel()), orig(t, *a, **k))[1]
    n = ddp.sync_bn_stats_(mods, 'mean')
    dist.all_reduce = orig
    assert n == 4 and calls == [4 + 4 + 5 + 5]                   # one collective over the concatenated buffers
    mean = [b.clone() for b in ddp.bn_buffers(mods)]
    for b, l in zip(ddp.bn_buffers(mods), local):                # restore, then the rank-0 mode
        b.copy_(l)
    ddp.sync_bn_stats_(mods, 'rank0')
    r0 = [b.clone() for b in ddp.bn_buffers(mods)]
    gathered = [[torch.empty_like(l) for _ in range(world)] for l in local]
    for l, gl in zip(local, gathered):
        dist.all_gather(gl, l)
    assert int(mods[1].num_batches_tracked) == 1
    if rank == 0:
        torch.save({'locals': gathered, 'mean': mean, 'rank0': r0}, out)
    else:
        torch.save({'mean': mean, 'rank0': r0}, out + '.r1')
    dist.destroy_process_group()


def test_bn_running_stats_sync_two_ranks(tmp_path):
    out = str(tmp_path / 'bn.pt')
    mp.spawn(_bn_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out), torch.load(out + '.r1')
    for (l0, l1), m0, m1, a0, a1 in zip(r0['locals'], r0['mean'], r1['mean'], r0['rank0'], r1['rank0']):
        assert not torch.equal(l0, l1)                            # the ranks really diverged
        assert torch.allclose(m0, (l0 + l1) / 2, rtol=1e-6, atol=1e-9) and torch.equal(m0, m1)
        assert torch.equal(a0, l0) and torch.equal(a1, l0)


def _sparse_empty_worker(rank, world, port, out):
    """A rank with NOTHING pending (count 0, a 1-entry dummy list) still takes part in every collective of the sparse row exchange."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ha2g_amd import ddp
    C = 6
    if rank == 0:
        ids = torch.tensor([0, 7, 3, 9], dtype=torch.int64)
        rows = torch.arange(4 * C, dtype=torch.float32).view(4, C)
        cnt = torch.tensor([4], dtype=torch.int32)
    else:
        ids, rows, cnt = torch.zeros(1, dtype=torch.int64), torch.zeros(1, C), torch.zeros(1, dtype=torch.int32)
    ids_all, rows_all = ddp.gather_sparse_rows(ids, cnt, rows)
    assert ids_all.shape == (8,) and rows_all.shape == (8, C)
    assert ids_all[:4].tolist() == [0, 7, 3, 9] and int(ids_all[4:].abs().sum()) == 0 and float(rows_all[4:].abs().sum()) == 0.0
    assert torch.allclose(rows_all[:4], torch.arange(4 * C, dtype=torch.float32).view(4, C) * 0.5)
    # both ranks empty: still one (dummy) entry per rank travels, never an empty collective
    e = ddp.gather_sparse_rows(torch.zeros(1, dtype=torch.int64), torch.zeros(1, dtype=torch.int32), torch.zeros(1, C))
    assert e[0].shape == (2,) and float(e[1].abs().sum()) == 0.0
    if rank == 0:
        torch.save({'ok': True}, out)
    dist.destroy_process_group()


def test_sparse_row_gather_with_an_idle_rank(tmp_path):
    out = str(tmp_path / 'spe.pt')
    mp.spawn(_sparse_empty_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert torch.load(out)['ok']


def _flag_worker(rank, world, port, out):
    """Round 4: ddp.sync_flag_ (MAX of the cluster-GRU error word: every replica skips a flagged step, or none does) and the touched-row count that is
    exchanged one step ahead (ddp.prefetch_max_count -> gather_sparse_rows(max_count=...): same result as the count exchange inside the gather)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ha2g_amd import ddp
    res = {}
    for name, mine in (('none', 0), ('rank1_only', 7 if rank == 1 else 0), ('both', 3 + rank)):
        flag = torch.tensor([mine], dtype=torch.int32)
        ddp.sync_flag_(flag)
        res[name] = int(flag.item())
    ddp.sync_flag_(None)                                     # no error word on this device: a no-op, not a collective
    # prefetched maximum of the per-rank row counts, then the gather sized by it
    cap, C = 9, 4
    g = torch.Generator().manual_seed(5 + rank)
    n = 3 if rank == 0 else 6
    ids = torch.zeros(cap, dtype=torch.int64); ids[:n] = torch.randperm(50, generator=g)[:n] + 1
    rows = torch.zeros(cap, C); rows[:n] = torch.randn(n, C, generator=g)
    count = torch.tensor(n, dtype=torch.int32)
    h = ddp.prefetch_max_count(count)
    res['max_count'] = h.value()
    a_ids, a_rows = ddp.gather_sparse_rows(ids, count, rows, max_count=h)
    b_ids, b_rows = ddp.gather_sparse_rows(ids, count, rows)
    res['same'] = bool(torch.equal(a_ids, b_ids) and torch.equal(a_rows, b_rows))
    res['shape'] = tuple(a_ids.shape)
    torch.save(res, out + '.r%d' % rank)
    dist.destroy_process_group()


def test_error_flag_and_prefetched_row_count_two_ranks(tmp_path):
    out = str(tmp_path / 'flag.pt')
    mp.spawn(_flag_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + '.r0'), torch.load(out + '.r1')
    assert r0 == r1                                          # both replicas see the same flags and the same gathered rows
    assert r0['none'] == 0 and r0['rank1_only'] == 7 and r0['both'] == 4
    assert r0['max_count'] == 6 and r0['shape'] == (12,) and r0['same']


class _FakeTable:
    """what ddp.exchange_sparse_ touches of ops.SparseTable (the real one needs the device for its compaction kernels)"""

    def __init__(self, C):
        self.weight = torch.zeros(50, C)
        self.map = None
        self.pending, self.count_hint, self.n_prefetch = [], None, 0

    def merged(self):
        assert len(self.pending) == 1
        return self.pending[0]


def _exchange_worker(rank, world, port, out):
    """ddp.exchange_sparse_ with a prefetched count AND an idle rank.  Rank 1 ran the forward (it holds the prefetched handle) but has
    nothing pending (a skipped backward): both ranks must take the SAME sequence of collectives -- the handle path, two all-gathers -- or the
    1-element count all-gather of one rank pairs with the id all-gather of the other (hang / size mismatch; this worker has a timeout)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ha2g_amd import ddp, ops
    calls = []
    orig_ag = ddp.all_gather_
    ddp.all_gather_ = lambda outs, t, group=None: (calls.append(tuple(t.shape)), orig_ag(outs, t, group))[1]
    ops.merge_rows = lambda ids, rows, m: (ids, torch.tensor([ids.numel()], dtype=torch.int32), rows)      # identity stand-in for the device compaction
    C, cap = 4, 9
    res = {}
    for case, n_prefetch in (('hint', 1), ('two_forwards', 2), ('mismatch', 1)):
        tb = _FakeTable(C)
        n = 5 if rank == 0 else 3
        ids = torch.zeros(cap, dtype=torch.int64); ids[:n] = torch.arange(1, n + 1) + 10 * rank
        rows = torch.zeros(cap, C); rows[:n] = 1.0 + rank
        count = torch.tensor(n, dtype=torch.int32)
        for _ in range(n_prefetch):                             # every rank runs every forward: the count collectives are symmetric
            tb.count_hint = (count, ddp.prefetch_max_count(count))
            tb.n_prefetch += 1
        if rank == 0:
            tb.pending = [(ids, count, rows)]                   # rank 1: idle (nothing pending), but it holds the handle
        if case == 'mismatch' and rank == 0:
            # rank 0's pending gradient is NOT the prefetched forward's (another count tensor); rank 1 is idle and cannot see that.
            # Nobody raises alone inside the collectives: both complete this exchange and BOTH raise at the table's next one.
            tb.pending = [(ids, count.clone(), rows)]
        calls.clear()
        ddp.exchange_sparse_(tb)
        res[case] = dict(calls=list(calls), n_ids=int(tb.pending[0][0].numel()), state=(tb.count_hint, tb.n_prefetch),
                         rows_sum=float(tb.pending[0][2].sum()))
        if case == 'mismatch':
            tb.pending = []
            try:
                ddp.exchange_sparse_(tb)
                res['mismatch_raised'] = False
            except RuntimeError as e:
                res['mismatch_raised'] = 'dropped rows' in str(e)
    torch.save(res, out + '.r%d' % rank)
    dist.destroy_process_group()


def test_sparse_exchange_with_prefetched_count_and_an_idle_rank(tmp_path):
    out = str(tmp_path / 'xchg.pt')
    mp.spawn(_exchange_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + '.r0'), torch.load(out + '.r1')
    assert r0 == r1                                              # same collectives, same merged result on both ranks
    assert [len(c) for c in r0['hint']['calls']] == [1, 2]       # ids [mx + 1 flag word], rows [mx, C]: NO count all-gather on either rank
    assert r0['hint']['calls'][0] == (6,) and r0['hint']['n_ids'] == 10 and r0['hint']['rows_sum'] == 5 * 4 * 0.5
    assert r0['mismatch']['calls'] == r0['hint']['calls'] and r0['mismatch_raised'] is True and r1['mismatch_raised'] is True
    assert r0['two_forwards']['calls'][0] == (1,) and len(r0['two_forwards']['calls']) == 3     # in-line count exchange everywhere
    assert r0['hint']['state'] == (None, 0)

"""Data-parallel helpers (one process per GPU; backend "nccl" = RCCL over xGMI on ROCm, "gloo" in CPU tests).

The train step shards by sample: every rank holds a full replica and its own B-sample batch; the only
exchange per step is the average of the flat per-module gradient buffers.  The helpers are device-agnostic so that the
world_size-2 gloo test exercises the same code path the GPUs run.
"""
import torch
import torch.distributed as dist


FORCE_ACTIVE = False      # tests: take the collective code path even in a world of one rank (the arithmetic is then the identity)
DISABLED = False          # tests: a rank of an initialised world computes its purely local step (reference value of the averaged one)


def active(group=None):
    if DISABLED or not (dist.is_available() and dist.is_initialized()):
        return False
    return FORCE_ACTIVE or dist.get_world_size(group) > 1


class _Done:
    """Work handle of a collective that was completed synchronously (host-staged gloo path)."""

    def wait(self):
        return True


def _host_staged(t, group):
    """gloo with device tensors (the world-size-2 hardware test: two processes on ONE GPU, where RCCL refuses duplicate devices): gloo's
    device-tensor support differs per collective and per build, so device tensors are staged through the host.  Never taken under RCCL."""
    return t.is_cuda and dist.get_backend(group) == 'gloo'


def all_reduce_(t, group=None, async_op=False):
    if _host_staged(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
        return _Done() if async_op else None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def sync_flag_(flag, group=None):
    """MAX over ranks of a device int32 flag word (the cluster-GRU error word that guards the optimizer kernels, FusedAdam.step): either every
    replica applies a step or none does -- a rank-local time-out must not let the replicas diverge.  One 4-byte collective."""
    if flag is None or not active(group):
        return
    if _host_staged(flag, group):
        h = flag.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX, group=group)
        flag.copy_(h)
        return
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)


def all_gather_(outs, t, group=None):
    if _host_staged(t, group):
        hs = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
        dist.all_gather(hs, t.cpu(), group=group)
        for o, h in zip(outs, hs):
            o.copy_(h)
        return
    dist.all_gather(outs, t, group=group)


def broadcast_one_(t, src=0, group=None):
    if _host_staged(t, group):
        h = t.cpu()
        dist.broadcast(h, src, group=group)
        t.copy_(h)
        return
    dist.broadcast(t, src, group=group)


def average_(flat, group=None, async_op=False):
    """In-place mean over ranks of a flat gradient buffer: pre-scale by 1/world then SUM all-reduce (one collective)."""
    ws = dist.get_world_size(group)
    if ws == 1 and not FORCE_ACTIVE:
        return None
    flat.mul_(1.0 / ws)
    return all_reduce_(flat, group, async_op)


def broadcast_(tensors, src=0, group=None):
    for t in tensors:
        broadcast_one_(t, src, group)


def bn_buffers(modules):
    """The BatchNorm running statistics (running_mean / running_var buffers) of a list of modules, in registration order."""
    return [b for m in modules for k, b in m.named_buffers() if k.endswith(('running_mean', 'running_var'))]


def sync_bn_stats_(modules, mode='mean', group=None):
    """BatchNorm running statistics are rank-local under data parallelism (each rank normalises its own shard).  Before a checkpoint
    or an evaluation pass make them identical on every rank: mode 'mean' = average over ranks (ONE all-reduce over the concatenated buffers),
    'rank0' = adopt rank 0's (what torch DDP's broadcast_buffers does).  num_batches_tracked is equal on every rank by construction."""
    bufs = bn_buffers(modules)
    if not bufs or not active(group):
        return 0
    flat = torch.cat([b.reshape(-1) for b in bufs])
    if mode == 'mean':
        average_(flat, group)
    elif mode == 'rank0':
        broadcast_one_(flat, 0, group)
    else:
        raise ValueError('sync_bn_stats_: mode must be "mean" or "rank0", got %r' % (mode,))
    o = 0
    for b in bufs:
        b.copy_(flat[o:o + b.numel()].view_as(b))
        o += b.numel()
    return len(bufs)


def rank_seed(base, rank=None):
    """Per-rank data / dropout seed: ranks must draw different batches and masks (weak scaling)."""
    if rank is None:
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    return base + rank


class _CountHandle:
    """max over ranks of a device-side row count, on its way to the host (prefetch_max_count)"""

    def __init__(self, host, event):
        self.host, self.event = host, event

    def value(self):
        if self.event is not None:
            self.event.synchronize()                       # fired during the forward: no stall in practice
        return max(int(self.host.max().item()), 1)


def prefetch_max_count(count, group=None):
    """Enqueue (do not wait for) the all-gather of one int32 device count over the ranks and the copy of the result to pinned host memory."""
    ws = dist.get_world_size(group)
    cnt = count.to(torch.int64).reshape(1)
    if _host_staged(cnt, group) or not cnt.is_cuda:
        counts = [torch.empty(1, dtype=torch.int64) for _ in range(ws)]
        dist.all_gather(counts, cnt.cpu(), group=group)
        return _CountHandle(torch.stack(counts).reshape(-1), None)
    out = torch.empty(ws, dtype=torch.int64, device=cnt.device)
    dist.all_gather_into_tensor(out, cnt, group=group)       # stream-ordered on the compute stream (RCCL): no host wait
    host = torch.empty(ws, dtype=torch.int64, pin_memory=True)
    host.copy_(out, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return _CountHandle(host, ev)


def gather_sparse_rows(ids, count, rows, group=None, max_count=None, flag=None):
    """All-gather of compact row gradients: every rank contributes (ids [cap], device count, rows [cap, C]) of which the first
    `count` entries are valid.  Only max-over-ranks(count) rows travel (one tiny count all-gather + host read decides the size);
    entries past a rank's count come back as id 0 with a zero row, rows are pre-scaled by 1/world (mean).  -> (ids [W*mx], rows [W*mx, C]).
    flag (an int, 0 = fine): a rank-local error word that rides as one extra element of the id all-gather -- no collective of its own -- and
    comes back MAX-reduced over the ranks as a third result (a 0-d tensor on the ids' device): every rank learns that SOME rank was in trouble.
    Device-agnostic (RCCL on the GPUs, gloo in the CPU tests)."""
    ws = dist.get_world_size(group)
    cnt = count.to(torch.int64).reshape(1)
    if max_count is not None:                              # prefetched at forward time (ops.SparseTable.prefetch_count): no host stall here
        mx = max_count.value()
    else:
        counts = [torch.empty_like(cnt) for _ in range(ws)]
        all_gather_(counts, cnt, group)
        mx = max(int(torch.stack(counts).max().item()), 1)   # never an empty collective
    if ids.shape[0] < mx:                                  # a rank whose own list is shorter than the longest one (e.g. nothing pending)
        ids = torch.cat([ids, ids.new_zeros(mx - ids.shape[0])])
        rows = torch.cat([rows, rows.new_zeros(mx - rows.shape[0], rows.shape[1])])
    valid = torch.arange(mx, device=ids.device) < cnt
    ids_s = torch.where(valid, ids[:mx], torch.zeros_like(ids[:mx])).contiguous()
    if flag is not None:
        ids_s = torch.cat([ids_s, ids_s.new_full((1,), int(flag))])
    rows_s = (rows[:mx] * valid.unsqueeze(1) * (1.0 / ws)).contiguous()
    ids_all = [torch.empty_like(ids_s) for _ in range(ws)]
    rows_all = [torch.empty_like(rows_s) for _ in range(ws)]
    all_gather_(ids_all, ids_s, group)
    all_gather_(rows_all, rows_s, group)
    if flag is not None:
        worst = torch.stack([t[-1] for t in ids_all]).max()
        return torch.cat([t[:-1] for t in ids_all]), torch.cat(rows_all), worst
    return torch.cat(ids_all), torch.cat(rows_all)


_MISMATCH_MSG = ('ha2g_amd.ddp.exchange_sparse_: on some rank one forward was prefetched but the pending gradient was not that forward\'s: the '
                 'prefetched row count did not bound it and the previous exchange of this table dropped rows')


def _raise_if_flagged(table):
    """The flag of the PREVIOUS exchange of this table (MAX over ranks, so identical everywhere), looked at with a lag of one exchange: by then its
    copy has long landed, and every rank raises at the same call -- none is left waiting inside a collective for a peer that raised alone."""
    w = getattr(table, '_xflag', None)
    if w is None:
        return
    table._xflag = None
    ev, word = w
    if ev is not None:
        ev.synchronize()
    if int(word.reshape(-1)[0]) != 0:
        raise RuntimeError(_MISMATCH_MSG)


def exchange_sparse_(table, group=None):
    """Data-parallel mean of one sparse table's compact gradient: gather every rank's rows, merge repeated ids (deterministic row
    sums in rank order) -- replaces the dense n_words x 300 all-reduce."""
    if not active(group):
        return
    from . import ops
    # EVERY rank takes part in the same sequence of collectives, whatever its local state: a rank with nothing pending (a skipped backward, a
    # non-finite-loss skip on that rank only) contributes count 0 -- an early return here would leave the other ranks' collectives hanging.
    # Whether the prefetched max count replaces the in-line count all-gather is decided from RANK-SYMMETRIC state only: exactly one count
    # collective was issued since the last exchange (every rank ran that one forward, so every rank holds its handle -- with or without a
    # pending gradient: the handle's maximum bounds every rank's count of that forward).  Anything else takes the in-line exchange everywhere.
    _raise_if_flagged(table)
    hint = table.count_hint[1] if (table.n_prefetch == 1 and table.count_hint is not None) else None
    bad = 0
    if table.pending:
        if hint is not None and not (len(table.pending) == 1 and table.count_hint[0] is table.pending[0][1]):
            # one forward was prefetched but the pending gradient is not that forward's: the prefetched count does not bound it.  `table.pending` is
            # RANK-LOCAL (a peer whose backward was skipped holds nothing and sees no mismatch), so raising here would leave the peers waiting in
            # the all-gathers below.  The collectives are completed -- rows past the hint are dropped -- with the error riding in the id
            # all-gather, and EVERY rank raises together at this table's next exchange (_raise_if_flagged).
            bad = 1
        ids, count, rows = table.merged()
    else:
        w = table.weight
        ids = torch.zeros(1, dtype=torch.int64, device=w.device)
        count = torch.zeros(1, dtype=torch.int32, device=w.device)
        rows = torch.zeros(1, w.shape[1], dtype=torch.float32, device=w.device)
    ids_all, rows_all, worst = gather_sparse_rows(ids, count, rows, group, max_count=hint, flag=bad)
    if worst.is_cuda:
        word = torch.empty(1, dtype=torch.int64, pin_memory=True)
        word.copy_(worst.reshape(1), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        table._xflag = (ev, word)
    else:
        table._xflag = (None, worst)
    table.count_hint, table.n_prefetch = None, 0
    table.pending = [ops.merge_rows(ids_all, rows_all, table.map)] if ids_all.numel() else []


def average_module_grads_(optimizers, group=None):
    """Average gradients of every optimizer's parameters: one collective per FusedAdam (flat buffer), else per tensor."""
    if not active(group):
        return
    for o in optimizers:
        for tb in getattr(o, 'sparse_tables', ()):
            exchange_sparse_(tb, group)
        if hasattr(o, 'flat_g'):
            average_(o.flat_g, group)
        else:
            for grp in o.param_groups:
                for p in grp['params']:
                    if p.grad is not None:
                        average_(p.grad, group)

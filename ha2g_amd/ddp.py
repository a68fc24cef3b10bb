"""Data-parallel helpers (one process per GPU; backend "nccl" = RCCL over xGMI on ROCm, "gloo" in CPU tests).

The train step shards by sample: every rank holds a full replica and its own B-sample batch; the only
exchange per step is the average of the flat per-module gradient buffers.  The helpers are device-agnostic so that the
world_size-2 gloo test exercises the same code path the GPUs run.
"""
import torch
import torch.distributed as dist


FORCE_ACTIVE = False      # tests: take the collective code path even in a world of one rank (the arithmetic is then the identity)


def active(group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return FORCE_ACTIVE or dist.get_world_size(group) > 1


def average_(flat, group=None, async_op=False):
    """In-place mean over ranks of a flat gradient buffer: pre-scale by 1/world then SUM all-reduce (one collective)."""
    ws = dist.get_world_size(group)
    if ws == 1 and not FORCE_ACTIVE:
        return None
    flat.mul_(1.0 / ws)
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def broadcast_(tensors, src=0, group=None):
    for t in tensors:
        dist.broadcast(t, src, group=group)


def rank_seed(base, rank=None):
    """Per-rank data / dropout seed: ranks must draw different batches and masks (weak scaling)."""
    if rank is None:
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    return base + rank


def gather_sparse_rows(ids, count, rows, group=None):
    """All-gather of compact row gradients: every rank contributes (ids [cap], device count, rows [cap, C]) of which the first
    `count` entries are valid.  Only max-over-ranks(count) rows travel (one tiny count all-gather + host read decides the size);
    entries past a rank's count come back as id 0 with a zero row, rows are pre-scaled by 1/world (mean).  -> (ids [W*mx], rows [W*mx, C]).
    Device-agnostic (RCCL on the GPUs, gloo in the CPU tests)."""
    ws = dist.get_world_size(group)
    cnt = count.to(torch.int64).reshape(1)
    counts = [torch.empty_like(cnt) for _ in range(ws)]
    dist.all_gather(counts, cnt, group=group)
    mx = int(torch.stack(counts).max().item())
    valid = torch.arange(mx, device=ids.device) < cnt
    ids_s = torch.where(valid, ids[:mx], torch.zeros_like(ids[:mx])).contiguous()
    rows_s = (rows[:mx] * valid.unsqueeze(1) * (1.0 / ws)).contiguous()
    ids_all = [torch.empty_like(ids_s) for _ in range(ws)]
    rows_all = [torch.empty_like(rows_s) for _ in range(ws)]
    dist.all_gather(ids_all, ids_s, group=group)
    dist.all_gather(rows_all, rows_s, group=group)
    return torch.cat(ids_all), torch.cat(rows_all)


def exchange_sparse_(table, group=None):
    """Data-parallel mean of one sparse table's compact gradient: gather every rank's rows, merge repeated ids (deterministic row
    sums in rank order) -- replaces the dense n_words x 300 all-reduce."""
    if not active(group) or not table.pending:
        return
    from . import ops
    ids, count, rows = table.merged()
    ids_all, rows_all = gather_sparse_rows(ids, count, rows, group)
    table.pending = [ops.merge_rows(ids_all, rows_all, table.map)]


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

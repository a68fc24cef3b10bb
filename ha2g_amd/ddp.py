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


def average_module_grads_(optimizers, group=None):
    """Average gradients of every optimizer's parameters: one collective per FusedAdam (flat buffer), else per tensor."""
    if not active(group):
        return
    for o in optimizers:
        if hasattr(o, 'flat_g'):
            average_(o.flat_g, group)
        else:
            for grp in o.param_groups:
                for p in grp['params']:
                    if p.grad is not None:
                        average_(p.grad, group)

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

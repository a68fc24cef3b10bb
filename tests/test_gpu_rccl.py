"""RCCL smoke on the GPU box: a single-rank "nccl" process group (the backend the multi-GPU bench uses) must broadcast and
all-reduce the flat parameter / gradient buffers of a trainer and leave a train step working.  Runs in a child process so
the process group never leaks into the pytest process.  (The world_size-2 arithmetic is covered on CPU by test_ddp_gloo.)"""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(root)r); sys.path.insert(1, %(root)r + '/tests')
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
from bench import Vocab
from ha2g_amd import procedural as proc
from ha2g_amd.config import hierarchy_args
from ha2g_amd.train import HierarchyTrainer
args = hierarchy_args(hidden_size=32, n_layers=2)
tr = HierarchyTrainer(args, Vocab(50), Vocab(7), 27, dev)
tr.broadcast_parameters(0)
text, spec, target, vid = (torch.from_numpy(x).to(dev) for x in proc.make_batch(3, 27, 50, 7, 5))
r0 = tr.train_iter(11, text, spec, target, vid)
for o in tr.gen_opts + [tr.audio_opt, tr.text_opt, tr.dis_opt]:
    before = o.flat_g.clone()
    dist.all_reduce(o.flat_g)                      # world of one: must be the identity, through RCCL
    assert torch.equal(before, o.flat_g)
r1 = tr.train_iter(11, text, spec, target, vid)
torch.cuda.synchronize()
assert all(v == v for v in r1.values()), r1
dist.destroy_process_group()
print('RCCL_OK')
'''


CHILD_FORCED = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(root)r); sys.path.insert(1, %(root)r + '/tests')
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
from bench import Vocab
from ha2g_amd import ddp, ops, procedural as proc, train_hierarchy as th
from ha2g_amd.config import hierarchy_args
from ha2g_testing import no_dropout
from ha2g_amd.train import HierarchyTrainer
B = 3
text, spec, target, vid = (torch.from_numpy(x).to(dev) for x in proc.make_batch(B, 27, 50, 7, 5))
eps_const = torch.from_numpy(proc.tensor_for('in.eps', (3 * B, 16), 11)).to(dev)
perm = torch.from_numpy(proc.fixed_perm(B, 11)).to(dev)
th.randperm_source = lambda n, device: perm
calls = []
orig_all_reduce = dist.all_reduce
def counting_all_reduce(t, *a, **k):
    calls.append((t.numel(), bool(k.get('async_op'))))
    return orig_all_reduce(t, *a, **k)
dist.all_reduce = counting_all_reduce
def run(forced):
    torch.manual_seed(5)
    ops.rng.seed(dev, 77)
    tr = HierarchyTrainer(hierarchy_args(hidden_size=32, n_layers=2), Vocab(50), Vocab(7), 27, dev)
    for m in tr.modules():
        no_dropout(m)
    for g in tr.gens:
        g.eps_source = lambda shape, device: eps_const[:shape[0]]
    ddp.FORCE_ACTIVE = forced
    if forced:
        tr.broadcast_parameters(0)
    rets = [tr.train_iter(e, text, spec, target, vid) for e in (0, 11, 11)]
    ddp.FORCE_ACTIVE = False
    torch.cuda.synchronize()
    flat = torch.cat([o.flat_p for o in tr.gen_opts + [tr.audio_opt, tr.text_opt, tr.dis_opt]])
    return rets, flat.clone()
r_plain, p_plain = run(False)
assert not calls, calls
r_ddp, p_ddp = run(True)
# the in-step collective branch ran: per step 4 async all-reduces (the three generators after backward stage 1, the stand-alone text encoder
# after its own backward -- all in flight under the audio tower's backward) + the audio encoder's, and one for the discriminator in each
# GAN-phase step
assert len(calls) == 3 * 5 + 2, calls
assert sum(1 for _, a in calls if a) == 12, calls
assert r_plain == r_ddp, (r_plain, r_ddp)
assert torch.equal(p_plain, p_ddp)                 # a world of one: averaging is the identity, bit for bit
dist.destroy_process_group()
print('RCCL_FORCED_OK')
'''


CHILD_GRAPH = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(root)r); sys.path.insert(1, %(root)r + '/tests')
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
from bench import Vocab
from ha2g_amd import ddp, ops, procedural as proc, train_hierarchy as th
from ha2g_amd.config import hierarchy_args
from ha2g_testing import no_dropout
from ha2g_amd.train import HierarchyTrainer
B = 3
text, spec, target, vid = (torch.from_numpy(x).to(dev) for x in proc.make_batch(B, 27, 50, 7, 5))
eps_const = torch.from_numpy(proc.tensor_for('in.eps', (3 * B, 16), 11)).to(dev)
perm = torch.from_numpy(proc.fixed_perm(B, 11)).to(dev)
th.randperm_source = lambda n, device: perm
ddp.FORCE_ACTIVE = True                            # the in-step collectives run (a world of one: the identity, through RCCL)
def make():
    torch.manual_seed(5)
    ops.rng.seed(dev, 77)
    tr = HierarchyTrainer(hierarchy_args(hidden_size=32, n_layers=2), Vocab(50), Vocab(7), 27, dev, sparse_embeddings=False)
    for m in tr.modules():
        no_dropout(m)
    for g in tr.gens:
        g.eps_source = lambda shape, device: eps_const[:shape[0]]
    tr.broadcast_parameters(0)
    return tr
tr_e = make()
eager = []
for _ in range(5):
    names, packed = tr_e.train_iter(11, text, spec, target, vid, return_tensors=True)
    eager.append(packed.clone())
torch.cuda.synchronize()
tr_g = make()
graph, gnames, gpacked = tr_g.capture_step(11, text, spec, target, vid)     # 2 warm-up steps, then ONE captured step WITH its RCCL all-reduces
replays = []
for _ in range(2):
    graph.replay()
    replays.append(gpacked.clone())
torch.cuda.synchronize()
assert gnames == names
for got, ref in zip(replays, eager[2:4]):
    assert torch.equal(got, ref), (got.tolist(), ref.tolist())
pe = torch.cat([o.flat_p for o in tr_e.gen_opts + [tr_e.audio_opt, tr_e.text_opt, tr_e.dis_opt]])
tr_e2 = make()
for _ in range(4):
    tr_e2.train_iter(11, text, spec, target, vid, return_tensors=True)
torch.cuda.synchronize()
pg = torch.cat([o.flat_p for o in tr_g.gen_opts + [tr_g.audio_opt, tr_g.text_opt, tr_g.dis_opt]])
p4 = torch.cat([o.flat_p for o in tr_e2.gen_opts + [tr_e2.audio_opt, tr_e2.text_opt, tr_e2.dis_opt]])
assert torch.equal(pg, p4)                         # 2 warm-up + 2 replayed steps == 4 eager steps, every parameter, bit for bit
ddp.FORCE_ACTIVE = False
dist.destroy_process_group()
print('RCCL_GRAPH_OK')
'''


def _run_child(code):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    return subprocess.run([sys.executable, '-c', code % dict(root=ROOT)], env=env, capture_output=True, text=True, timeout=420)


def test_in_step_collective_branch_equals_plain_step():
    """Forces the data-parallel branch of the train step (async RCCL all-reduce of the three generators' flat gradient
    buffers issued after backward stage 1, stage-2 backward of the encoders, the encoders' all-reduces, wait, Adam; the
    discriminator's all-reduce inside the D phase) in a world of ONE rank and checks three steps equal the plain path bit
    for bit (loss dicts and every parameter)."""
    p = _run_child(CHILD_FORCED)
    assert p.returncode == 0 and 'RCCL_FORCED_OK' in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])


def test_single_rank_rccl_group():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, '-c', CHILD % dict(root=ROOT)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and 'RCCL_OK' in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])


def test_c_abi_allreduce_bucket_world_size_1():
    """The C-ABI's own collective entry points (csrc/comm.hip, for hosts without torch.distributed): id -> communicator -> in-place mean /
    sum all-reduce of a flat gradient bucket on the caller's stream -> destroy.  One rank is all a one-GPU box allows (RCCL refuses two
    ranks on one device); what is pinned here is the binding itself: RCCL resolved by dlopen, the handle round trip, stream ordering,
    and that a world of one leaves the bucket bit-identical under both reductions."""
    import ctypes
    import torch
    from ha2g_amd._lib import check, lib
    assert lib.ha2g_comm_available() == 1
    idb = ctypes.create_string_buffer(128)
    check(lib.ha2g_comm_unique_id(ctypes.addressof(idb)))
    assert any(idb.raw)
    comm = ctypes.c_void_p()
    torch.cuda.set_device(0)
    check(lib.ha2g_comm_init(ctypes.addressof(idb), 0, 1, ctypes.addressof(comm)))
    try:
        assert comm.value and lib.ha2g_comm_world(comm.value) == 1
        st = torch.cuda.Stream()
        x = torch.randn((1 << 20) + 3, device='cuda:0')
        ref = x.clone()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            x.mul_(2.0)                                                          # ordered before the collective on the same stream
            check(lib.ha2g_allreduce_bucket(comm.value, x.data_ptr(), x.numel(), 1, st.cuda_stream))
            check(lib.ha2g_allreduce_bucket(comm.value, x.data_ptr(), x.numel(), 0, st.cuda_stream))
            check(lib.ha2g_allreduce_bucket(comm.value, x.data_ptr(), 0, 0, st.cuda_stream))
        st.synchronize()
        assert torch.equal(x, ref * 2.0)
        assert lib.ha2g_allreduce_bucket(0, x.data_ptr(), 4, 0, st.cuda_stream) != 0    # a null communicator is an error, not a crash
    finally:
        check(lib.ha2g_comm_destroy(comm.value))


def test_whole_step_with_its_rccl_collectives_captures_into_a_hipgraph():
    """the data-parallel step -- D all-reduce inside the D phase, the generators' / text encoder's asynchronous all-reduces under
    the audio tower's backward, the audio bucket, the error-word MAX-reduce -- captured into ONE hipGraph in a world of one rank (ddp.FORCE_ACTIVE;
    dense embedding tables: the row-wise exchange reads a count on the host) and replayed: losses and every parameter equal the eager data-parallel
    steps bit for bit.  N > 1 ranks are then no longer bound to the eager host path (bench.py --launch graph).  Child process with a hard time limit:
    a collective inside a capture that hung would otherwise hang the suite."""
    p = _run_child(CHILD_GRAPH)
    assert p.returncode == 0 and 'RCCL_GRAPH_OK' in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])

"""Data-parallel train step with TWO ranks on hardware, and the cluster GRU beside foreign co-resident work.

No multi-GPU box is available to the test tier, and RCCL refuses two ranks on one device, so the world-size-2 step runs as two fresh child
processes that share cuda:0 and exchange through the gloo backend (device tensors staged through the host by ha2g_amd.ddp -- the same
collective call sites the RCCL run takes: D all-reduce inside the D phase, the generators' asynchronous all-reduces after backward stage 1,
the encoders' after stage 2, the sparse row gather).  Replaces the reference's DataParallel wrap (scripts/train.py:133-143).
"""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(root)r); sys.path.insert(1, %(root)r + '/tests')
rank = int(os.environ['RANK'])
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
dist.init_process_group('gloo', rank=rank, world_size=2)
from bench import Vocab
from ha2g_amd import ddp, ops, procedural as proc
from ha2g_amd.config import hierarchy_args
from ha2g_amd.train import HierarchyTrainer
ops.USE_GRU_CLUSTER = False                        # two processes on ONE GPU: two cluster launches cannot both be fully co-resident
B = 4
text, spec, target, vid = (torch.from_numpy(x).to(dev) for x in proc.make_batch(B, 27, 60, 9, ddp.rank_seed(500, rank)))

def make(sparse=False):
    torch.manual_seed(5)                           # identical replicas (broadcast_parameters makes it so in any case)
    return HierarchyTrainer(hierarchy_args(hidden_size=32, n_layers=2), Vocab(60), Vocab(9), 27, dev, sparse_embeddings=sparse)

def seed_draws():
    torch.manual_seed(100 + rank)                  # reparameterisation noise / randperm: rank-local
    ops.rng.seed(dev, ddp.rank_seed(1234, rank))   # dropout masks: rank-local

def grads(tr):                                     # generators + encoders (the D gradient of a warm-up step is zero)
    return torch.cat([o.flat_g for o in tr.gen_opts + [tr.audio_opt, tr.text_opt]]).clone()

def params(tr):
    return {'%%d.%%s' %% (i, k): p.detach().clone() for i, m in enumerate(tr.modules()) for k, p in m.named_parameters()}

def gather(t):
    outs = [torch.empty_like(t) for _ in range(2)]
    ddp.all_gather_(outs, t.contiguous())
    return outs

# 1. this rank's purely LOCAL gradient of a warm-up step (no D update before the G phase, so the G gradients of the local and the
#    data-parallel run start from the same parameters)
ddp.DISABLED = True
tr = make(); seed_draws()
assert not ddp.active()
r_loc = tr.train_iter(0, text, spec, target, vid)
torch.cuda.synchronize()
g_loc = grads(tr)
ddp.DISABLED = False
del tr

# 2. the data-parallel run, dense embedding gradients: 3 steps (warm-up, GAN, GAN)
assert ddp.active()
tr = make(); tr.broadcast_parameters(0); seed_draws()
r0 = tr.train_iter(0, text, spec, target, vid)
torch.cuda.synchronize()
g_ddp = grads(tr)
rets = [r0] + [tr.train_iter(11, text, spec, target, vid) for _ in range(2)]
tr.sync()
assert all(v == v and abs(v) < 1e6 for r in rets for v in r.values()), rets
assert r0 == r_loc, (r0, r_loc)                    # the forward of step 1 does not depend on the exchange
gl = gather(g_loc)
assert not torch.equal(gl[0], gl[1])               # the shards really differ
mean = (gl[0] + gl[1]) * 0.5
err = float((g_ddp - mean).abs().max() / mean.abs().max())
assert err < 1e-6, ('averaged gradient vs mean of the single-rank gradients', err)
p_dense = params(tr)
flat = torch.cat([p.reshape(-1) for p in p_dense.values()])
fl = gather(flat)
assert torch.equal(fl[0], fl[1]), 'replicas diverged (dense)'
losses = gather(torch.tensor([rets[-1]['loss']], device=dev))
assert float(losses[0]) != float(losses[1])        # different batches per rank
ops.rng.seed(dev, ddp.rank_seed(1234, rank))
masks = gather(ops.dropout_mask((4096,), 0.3, dev))
assert not torch.equal(masks[0], masks[1])         # different dropout masks per rank
# BatchNorm running statistics are rank-local during training; sync_bn_stats makes them identical (checkpoint / validation)
bn = torch.cat([b.reshape(-1) for b in ddp.bn_buffers(tr.modules())])
bl = gather(bn)
assert not torch.equal(bl[0], bl[1])
assert tr.sync_bn_stats('mean') == len(ddp.bn_buffers(tr.modules())) > 0
bn2 = torch.cat([b.reshape(-1) for b in ddp.bn_buffers(tr.modules())])
b2 = gather(bn2)
assert torch.equal(b2[0], b2[1]) and torch.allclose(b2[0], (bl[0] + bl[1]) * 0.5, rtol=1e-6, atol=1e-9)
del tr

# 3. the same three steps with row-wise (sparse) word-embedding tables: compact row gather instead of the dense table all-reduce
tr = make(sparse=True); tr.broadcast_parameters(0); seed_draws()
rets_s = [tr.train_iter(e, text, spec, target, vid) for e in (0, 11, 11)]
tr.sync(); tr.sync_sparse(); torch.cuda.synchronize()
p_sparse = params(tr)
flat_s = torch.cat([p.reshape(-1) for p in p_sparse.values()])
fs = gather(flat_s)
assert torch.equal(fs[0], fs[1]), 'replicas diverged (sparse tables)'
assert rets_s == rets, (rets_s, rets)
bad = [k for k in p_dense if not torch.equal(p_dense[k], p_sparse[k])]
assert not bad, ('sparse vs dense data-parallel parameters differ', bad[:5])
del tr

# 4. ADVICE r4: a cluster-GRU time-out on ONE rank is recovered from COLLECTIVELY (HierarchyTrainer._train_iter_ddp): the flag is MAX-reduced
#    inside the step, both replicas skip the flagged updates, both notice the flag at the same later call, restore their BatchNorm
#    statistics, switch to the fallback recurrences and go on -- no rank raises or re-runs a step on its own (unmatched collectives = hang)
tr = make(); tr.broadcast_parameters(0); seed_draws()
ops.USE_GRU_CLUSTER = True                         # H = 32 never takes the cluster kernels; the switch only records the fallback decision
word = ops._cluster_scratch(dev)[1]                # the error word exists on every rank (as it does after the first cluster launch)
def snap(tr):
    return (torch.cat([o.flat_p for o in tr.gen_opts + [tr.audio_opt, tr.text_opt, tr.dis_opt]]).clone(),
            torch.cat([b.reshape(-1).double() for b in tr._bn_buffers()]).clone(), [int(o.step_t.item()) for o in tr.gen_opts + [tr.dis_opt]])
for _ in range(2):
    tr.train_iter(11, text, spec, target, vid)     # calls 0, 1: clean
tr.sync()
p0, b0, s0 = snap(tr)
if rank == 1:
    word.fill_(1)                                  # as if a hand-off of THIS rank timed out during call 2
for _ in range(2):
    tr.train_iter(11, text, spec, target, vid)     # calls 2, 3: flagged on every rank after the in-step MAX-reduce -> optimizer no-ops
torch.cuda.synchronize()
assert tr.cluster_retries == 0                     # not noticed yet: the end-of-step words are examined with a fixed lag, the same on every rank
p1, b1, s1 = snap(tr)
assert torch.equal(p0, p1) and s0 == s1, 'a flagged data-parallel step changed parameters / step counters'
assert not torch.equal(b0, b1)                     # ... while its forwards did touch the BatchNorm statistics
assert int(word.item()) == 1                       # rank 0 holds the flag too: it came through ddp.sync_flag_
r4 = tr.train_iter(11, text, spec, target, vid)    # call 4: every rank recovers here and runs the batch
tr.sync()
assert tr.cluster_retries == 1 and not ops.USE_GRU_CLUSTER and int(word.item()) == 0
p2, b2, s2 = snap(tr)
assert s2 == [n + 1 for n in s0] and not torch.equal(p2, p0)          # exactly ONE update since the snapshot
assert all(v == v for v in r4.values())
fl = gather(p2)
assert torch.equal(fl[0], fl[1]), 'replicas diverged across the collective recovery'
# BatchNorm buffers: restored to the state before the first flagged call, then ONE step's updates (num_batches_tracked is among them)
ref_tr_cnt = [int(b) for b in tr._bn_buffers() if b.dtype == torch.int64]
tr2 = make(); tr2.broadcast_parameters(0)
for _ in range(3):
    tr2.train_iter(11, text, spec, target, vid)
tr2.sync()
assert ref_tr_cnt == [int(b) for b in tr2._bn_buffers() if b.dtype == torch.int64]     # as after three effective steps
tr2.train_iter(11, text, spec, target, vid); tr2.sync()
assert tr2.cluster_retries == 0
del tr, tr2

# 5. VERDICT r4 item 6c: the CLUSTER GRU on (H = 300: gru_cluster.hip forward + BPTT, 10 co-resident workgroups per launch and rank) under data
#    parallelism: two processes' cluster launches, the in-step collectives and the side-stream kernels share one device.  Each rank caps its launches
#    at half the device (ha2g_gru_cluster_tile_cap): what a process that shares its GPU must do.
from ha2g_amd._lib import lib
assert lib.ha2g_gru_cluster_supported(300)
lib.ha2g_gru_cluster_tile_cap(12)
ops.USE_GRU_CLUSTER = True
word.zero_(); torch.cuda.synchronize()
torch.manual_seed(5)
trc = HierarchyTrainer(hierarchy_args(), Vocab(60), Vocab(9), 27, dev)           # hidden_size 300, 4 layers: the cluster kernels
trc.broadcast_parameters(0); seed_draws()
rc = [trc.train_iter(e, text, spec, target, vid) for e in (0, 11, 11)]
trc.sync()
assert ops.gru_cluster_error(dev) == 0 and trc.cluster_retries == 0 and ops.USE_GRU_CLUSTER
assert all(v == v and abs(v) < 1e6 for r_ in rc for v in r_.values()), rc
flc = gather(torch.cat([o.flat_p for o in trc.gen_opts + [trc.audio_opt, trc.text_opt, trc.dis_opt]]))
assert torch.equal(flc[0], flc[1]), 'replicas diverged with the cluster GRU on'
lib.ha2g_gru_cluster_tile_cap(0)
dist.destroy_process_group()
print('DDP2_OK rank %%d' %% rank)
'''


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_step_on_one_gpu():
    """3 data-parallel steps, 2 ranks: replicas stay bit-identical (dense and sparse embedding tables, which also agree with each other bit for
    bit), the averaged gradient of step 1 equals the mean of the two single-rank gradients (1e-6), batches / dropout masks / BatchNorm running
    statistics are rank-local, sync_bn_stats() makes the latter identical; a cluster-GRU time-out injected on ONE rank is recovered from by BOTH
    ranks at the same call (flagged updates skipped everywhere, BatchNorm statistics restored, replicas still bit-identical)."""
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, '-c', WORKER % dict(root=ROOT)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=900))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and 'DDP2_OK rank %d' % r in so, (r, so[-2000:], se[-4000:])


SOAK = r'''
import sys, torch
sys.path.insert(0, %(root)r); sys.path.insert(1, %(root)r + '/tests')
dev = torch.device('cuda', 0)
from ha2g_amd import ops
from ha2g_amd._lib import lib, check
assert lib.ha2g_gru_cluster_supported(300)
H, T, B = 300, 34, 384                             # the fused 3-chain launch of the headline step: 24 tiles x 2 directions x 5 = 240 workgroups
torch.manual_seed(0)
x = torch.randn(B, T, 2 * H, device=dev) * 0.3
ws = []
for d in range(2):
    ws += [torch.randn(3 * H, 2 * H, device=dev) * 0.05, torch.randn(3 * H, H, device=dev) * 0.05, torch.randn(3 * H, device=dev) * 0.1,
           torch.randn(3 * H, device=dev) * 0.1]
ref = ops.bigru(x, ws, H)                          # undisturbed launch
torch.cuda.synchronize()
assert ops.gru_cluster_error(dev) == 0
sink = torch.zeros(1, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
status = []
for blocks, usec in ((16, 20000), (32, 20000), (64, 1500000)):
    with torch.cuda.stream(side):
        check(lib.ha2g_debug_occupy(blocks, usec, sink.data_ptr(), side.cuda_stream))
    y = ops.bigru(x, ws, H)                        # needs 240 co-resident workgroups while `blocks` compute units are held
    torch.cuda.synchronize()
    e = ops.gru_cluster_error(dev)
    if e == 0:
        assert torch.equal(y, ref), 'a completed cluster launch must be exact'
    status.append((blocks, usec, e))
    ops.gru_cluster_error_tensor(dev).zero_()
print('SOAK_OK', status)
'''


def test_cluster_gru_beside_foreign_resident_work_completes_or_flags():
    """The cluster GRU needs all of its workgroups co-resident (gru_cluster.hip).  With 16 / 32 / 64 compute units held by a foreign kernel on
    another stream (what an overlapping collective would do) a 240-workgroup launch must either complete with exact results (the held units
    were released in time, or enough were free) or terminate with its error word set -- the train step raises Ha2gClusterError on that word --
    and must never hang: the child process has a hard time limit."""
    p = subprocess.run([sys.executable, '-c', SOAK % dict(root=ROOT)], env=dict(os.environ), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and 'SOAK_OK' in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])
    print(p.stdout.strip().splitlines()[-1])


def test_bench_two_ranks_rehearsal_on_one_gpu():
    """`bench.py --gpus 2` end to end -- the command the driver runs on a multi-GPU node -- rehearsed on ONE GPU (HA2G_BENCH_REHEARSAL=1:
    both ranks on cuda:0, gloo instead of RCCL): the script spawns its two rank processes, broadcasts the parameters, runs warm-up, timed,
    roofline, warm-up-phase and exact-fp32 legs with the in-step collectives, takes the MAX over ranks and prints ONE JSON line from rank 0
    whose collective-observed world size is 2 and whose value is the sum over both ranks.  Not a measurement (the line says so)."""
    import json
    env = dict(os.environ, HA2G_BENCH_REHEARSAL='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '16',
                        '--n-words', '300', '--n-spk', '11', '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]                                   # rank 0 only
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['rccl_world'] == 2 and d['rehearsal'] is True and d['launch'] == 'eager'
    assert d['config']['global_batch'] == 32 and d['config']['parallelism'] == 'dp2'
    assert abs(d['value'] - 2 * 16 * 34 / (d['ms_per_step'] * 1e-3)) <= 1e-3 * d['value']
    for k in ('exact_fp32_matrix_core', 'warmup_phase', 'eager', 'roofline'):
        assert d[k] is not None
    assert all(v == v for v in d['last_step'].values())

#!/usr/bin/env python3
"""Headline benchmark: pose-frames/sec of the HA2G hierarchy train step (config/hierarchy.yml shapes, B=128 per GPU,
T=34, 27-d pose, fp32, GAN phase = epoch > loss_warmup) on N MI355X, synthetic data, random-init weights.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line (rank 0).  `value` = B*T*N / max-over-ranks step time; `roofline` = the bi-GRU forward kernel timed
with HIP events on its launch stream inside the timed region; `cpu_baseline` = the CPU oracle (a port of the reference's
step) timed on a bounded sample on this box's host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch


class Vocab:
    def __init__(self, n_words, weights=None):
        self.n_words = n_words
        self.word_embedding_weights = weights


def host_cpu():
    """(model name, physical cores of one socket) from /proc/cpuinfo; falls back to os.cpu_count()."""
    model, cores = 'unknown', set()
    try:
        phys = None
        for line in open('/proc/cpuinfo'):
            k, _, v = line.partition(':')
            k, v = k.strip(), v.strip()
            if k == 'model name':
                model = v
            elif k == 'physical id':
                phys = v
            elif k == 'core id' and phys is not None:
                cores.add((phys, v))
        sockets = {p_ for p_, _ in cores}
        n = len([c for c in cores if c[0] == min(sockets)]) if cores else 0
    except OSError:
        n = 0
    return model, (n or os.cpu_count() or 1)


def cpu_baseline(epoch, n_words, n_spk, budget_s=60.0):
    """The CPU oracle's train step (oracle/ha2g_oracle.py, parity-pinned port of the reference) timed on this box's host cores:
    threads = physical cores of one socket (SURVEY 8d / BASELINE.md 3).  B=4 (BASELINE config 1): 1 warm-up + 3 timed steps
    per phase; B=128 (the headline batch): 1 warm-up step at B=4 to spin up the thread pool, then up to 3 timed GAN-phase
    steps, bounded by `budget_s` of CPU time (one step is ~20-40 s)."""
    from ha2g_amd import procedural as proc, schema
    from ha2g_amd.config import hierarchy_args
    from oracle import ha2g_oracle as O
    model, ncores = host_cpu()
    torch.set_num_threads(ncores)
    args = hierarchy_args(dropout_prob=0.0)
    sch = schema.step_schema(schema.GESTURE_POSE_DIMS, n_words, n_spk, args.hidden_size, args.n_layers)
    sd = schema.procedural_state(sch, 3)
    tr = O.OracleTrainer(sd, args)
    eps = lambda shp: torch.randn(shp)

    def batch(B):
        return tuple(map(torch.from_numpy, proc.make_batch(B, 27, n_words, n_spk, 4321)))

    def timed(B, ep, nmax, budget):
        data, times = batch(B), []
        while len(times) < nmax and (not times or sum(times) + times[-1] < budget):
            t0 = time.time()
            tr.train_iter(ep, *data, eps, torch.randperm(B))
            times.append(time.time() - t0)
        return times

    tr.train_iter(epoch, *batch(4), eps, torch.randperm(4))               # warm-up (thread pool, allocator)
    b4 = {}
    for name, ep in (('gan_phase', epoch), ('warmup_phase', 0)):
        ts = timed(4, ep, 3, 30.0)
        b4[name] = dict(steps=len(ts), s_per_step=round(sum(ts) / len(ts), 3), value=round(4 * 34 * len(ts) / sum(ts), 1))
    ts = timed(128, epoch, 3, budget_s)
    dt = sum(ts) / len(ts)
    return dict(value=round(128 * 34 / dt, 1), unit='pose-frames/s', cores=ncores, cpu_model=model, kind='port',
                sample='%d timed train step(s) (epoch %d, GAN phase) of B=128, T=34 on the CPU oracle after a warm-up step: %.1f s/step; '
                       'B=4 (BASELINE config 1): 1 warm-up + 3 timed steps per phase' % (len(ts), epoch, dt),
                b128=dict(steps=len(ts), s_per_step=round(dt, 2)), b4=b4)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU, RCCL rendezvous on
    127.0.0.1) BEFORE this process touches the GPU, stream rank 0's JSON line through, exit with the worst return code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p_ in procs:
        rc = max(rc, abs(p_.wait()))
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=128, help='per-GPU batch (weak scaling)')
    ap.add_argument('--epoch', type=int, default=11, help='> loss_warmup (10) = GAN phase; 0 = warm-up phase')
    ap.add_argument('--n-words', type=int, default=20000)
    ap.add_argument('--n-spk', type=int, default=1371)
    ap.add_argument('--expressive', action='store_true', help='config_expressive/hierarchy.yml: 6 levels, 126-d pose (BASELINE config 3)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--bf16', action='store_true', help='BASELINE config 5 style: every vectorisable GEMM / convolution with plain bf16 operands (fp32 accumulate, fp32 storage / master weights); reported with dtype "bf16", never the default')
    ap.add_argument('--fp32-storage', action='store_true', help='with --bf16: keep the audio trunk\'s activations / activation gradients in fp32 (bf16 operands only, the round-2 form of the mode) -- the "before" leg of the storage A/B')
    ap.add_argument('--launch', choices=('auto', 'graph', 'eager'), default='auto', help='what `value` times.  auto = eager: EAGER launches are the documented default launch form of the step (the form every N uses; the hipGraph replay of the same step is timed beside it at N = 1 and printed as `graph_replay`, never folded into `value`); graph: `value` = hipGraph replays')
    ap.add_argument('--n-poses', type=int, default=34, help='frames per window.  34 = the reference (parity).  62 = the legal neighbour of BASELINE config 5\'s T = 64 (SURVEY M5: the three audio taps agree only for T = 2 mod 4; T = 64 cannot exist with the reference modules): spectrogram width 126, discriminator out2 = Linear(56, 1) -- a PERFORMANCE-ONLY leg, no reference twin')
    ap.add_argument('--max-cluster-retries', type=int, default=1, help='fail (exit 3, no JSON line) when more cluster-GRU recoveries than this happened during the run: each one costs a whole step and switches the process to the slower single-workgroup recurrences, so a line above the bound does not measure the default path')
    ap.add_argument('--graph', action='store_true', help='same as --launch graph')
    ap.add_argument('--no-roofline', action='store_true', help='with --primary-only: skip the per-launch HIP-event pass as well (profiling runs)')
    ap.add_argument('--sparse-embeddings', action='store_true', help='compact row gradients + lazy row-wise Adam for the word-embedding tables (bit-identical to the dense default; -45 %% gradient-exchange bytes under data parallelism, +0.4 ms of small kernels on one GPU)')
    ap.add_argument('--dense-embeddings', action='store_true', help='data parallel: keep the dense word-embedding gradient all-reduce (the default under N > 1 is the row-wise exchange)')
    ap.add_argument('--dry-run', action='store_true', help='only exercise the rank launch: every rank prints its RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* and exits (no GPU)')
    ap.add_argument('--host-profile', action='store_true', help='cProfile of `--steps` eager steps after the warm-up (where the host time of a step goes): prints the top functions and exits, no JSON line')
    ap.add_argument('--primary-only', action='store_true', help='skip the secondary timings (warm-up phase, exact-fp32 mode): for profiling')
    a = ap.parse_args()
    if a.graph:
        a.launch = 'graph'

    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        spawn_ranks(a.gpus)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if a.dry_run:
        print(json.dumps(dict(dry_run=True, rank=rank, local_rank=local, world=world, master='%s:%s' % (os.environ.get('MASTER_ADDR'), os.environ.get('MASTER_PORT')),
                              ipc_legacy=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'))), flush=True)
        return
    assert world == a.gpus, 'launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)' % (a.gpus, world)
    # HA2G_BENCH_REHEARSAL=1: all N ranks share cuda:0 and talk over gloo (device tensors staged through the host by ha2g_amd.ddp) -- a
    # rehearsal of this script's multi-rank control flow (rank spawn, broadcast, barriers, in-step collectives, MAX over ranks, rccl_world)
    # on a one-GPU box, where RCCL refuses two ranks on one device.  NOT a measurement: the line says `rehearsal: true`.
    rehearsal = os.environ.get('HA2G_BENCH_REHEARSAL') == '1'
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    import torch.distributed as dist
    if world > 1:
        if rehearsal:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)

    from ha2g_amd import ops, procedural as proc
    if rehearsal and world > 1:
        ops.USE_GRU_CLUSTER = False                       # cluster launches of two processes on one GPU cannot both be fully co-resident
    from ha2g_amd.config import hierarchy_args
    from ha2g_amd.train import HierarchyTrainer

    torch.manual_seed(0)
    from ha2g_amd import schema
    from ha2g_amd._lib import lib as _lib0
    from ha2g_amd._lib import DEFAULT_GEMM_MODE
    default_mode = int(os.environ['HA2G_GEMM_MODE']) if 'HA2G_GEMM_MODE' in os.environ else (22 if a.bf16 else DEFAULT_GEMM_MODE)
    _lib0.ha2g_gemm_set_mode(default_mode)
    bwd_pieces = _lib0.ha2g_gemm_bwd_pieces()                # 3 = fp32-class backward (the default), 2 = 16-bit operand mantissa, 0 = fp32 MFMA / bf16
    from ha2g_amd import wav_engine as _we
    b16_storage = bool(a.bf16 and not a.fp32_storage)
    _we.set_b16(b16_storage)                                  # BASELINE config 5: bf16 activation / activation-gradient storage in the audio trunk
    P = 126 if a.expressive else 27
    T = a.n_poses
    assert T % 4 == 2, '--n-poses must be 2 mod 4 (the three audio taps yield W/2-1, 2*ceil(W/4)-2, 4*ceil(W/8)-2 frames: equal only then; SURVEY M5)'
    spec_w = 2 * (T + 1)                                      # 34 -> 70, 62 -> 126
    args = hierarchy_args(expressive=a.expressive, n_poses=T) # config[_expressive]/hierarchy.yml, dropout 0.3
    tr = HierarchyTrainer(args, Vocab(a.n_words), Vocab(a.n_spk), P, dev,
                          pose_dims=schema.EXPRESSIVE_POSE_DIMS if a.expressive else schema.GESTURE_POSE_DIMS,
                          sparse_embeddings=True if a.sparse_embeddings else (False if a.dense_embeddings else None))
    if world > 1:
        tr.broadcast_parameters(0)
    ops.rng.seed(dev, 1234 + rank)
    text, spec, target, vid = (torch.from_numpy(x).to(dev) for x in proc.make_batch(a.batch, P, a.n_words, a.n_spk, 1234 + rank, T=T, W=spec_w))

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    rccl_world = 1
    if world > 1:                                    # what RCCL itself sees: SUM all-reduce of one 1 per rank
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        rccl_world = int(ones.item())

    from ha2g_amd import train_hierarchy as th
    from ha2g_amd.train_hierarchy import _ret_dict
    from ha2g_amd._lib import lib as _lib

    def timed_eager(epoch, steps, clock=None):
        """`steps` eager train steps between barriers + device syncs -> (seconds, last loss dict)."""
        last_ = None
        sync()
        th.host_clock = clock
        t0 = time.perf_counter()
        for _ in range(steps):
            last_ = tr.train_iter(epoch, text, spec, target, vid)
        sync()
        dt_ = time.perf_counter() - t0
        th.host_clock = None
        tr.sync()                                    # outside the timed region: raises if a cluster-GRU hand-off of any step timed out
        return dt_, last_

    def timed_graph(epoch, steps):
        """The whole step captured into ONE hipGraph (same kernels, both streams, fresh dropout / noise / Adam counters per replay),
        `steps` replays between barriers + device syncs -> (seconds, last loss dict).  The capture's memory pool is released afterwards."""
        graph, gnames, gpacked = tr.capture_step(epoch, text, spec, target, vid)
        graph.replay()                               # one untimed replay: first-launch upload of the executable graph
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            graph.replay()
        sync()
        dt_ = time.perf_counter() - t0
        last_ = _ret_dict(args, gnames, gpacked.tolist())
        del graph, gpacked
        torch.cuda.empty_cache()
        return dt_, last_

    for _ in range(a.warmup):
        tr.train_iter(a.epoch, text, spec, target, vid)
    if a.host_profile:
        import cProfile, pstats, inspect
        # host time per autograd Function (forward and backward; the backward runs on the engine's thread, which cProfile does not see)
        from ha2g_amd import hierarchy_net as _hn, wav_engine as _we
        acc, deep_prof = {}, {}

        def _wrap(cls, name):
            fn = getattr(cls, name)

            deep = name == 'backward' and cls.__name__ in ('WavEncoderFunction', 'BiGRUFunction')     # the engine thread's two biggest: own cProfile

            def timed_fn(*x, **k):
                t_ = time.perf_counter()
                pr_ = None
                if deep:
                    pr_ = deep_prof.setdefault(cls.__name__, cProfile.Profile())
                    pr_.enable()
                try:
                    return fn(*x, **k)
                finally:
                    if pr_ is not None:
                        pr_.disable()
                    e = acc.setdefault('%s.%s' % (cls.__name__, name), [0, 0.0])
                    e[0] += 1; e[1] += time.perf_counter() - t_
            setattr(cls, name, staticmethod(timed_fn))
        for mod in (ops, _hn, _we):
            for _, cls in inspect.getmembers(mod, inspect.isclass):
                if issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function and cls.__module__ == mod.__name__:
                    _wrap(cls, 'forward'); _wrap(cls, 'backward')
        sync()
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        pr.enable()
        for _ in range(a.steps):
            tr.train_iter(a.epoch, text, spec, target, vid)
        pr.disable()
        t1 = time.perf_counter()
        sync()
        print('host wall per step (python running under the profiler, GPU queued behind): %.1f ms; %d steps' % ((t1 - t0) / a.steps * 1e3, a.steps))
        print('host time inside autograd Functions (inclusive of nested Functions), ms per step:')
        for k_, (n_, t_) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:40]:
            print('  %-44s %6d calls/step %8.3f ms/step' % (k_, n_ // a.steps, t_ / a.steps * 1e3))
        for nm, p_ in deep_prof.items():
            print('---- inside %s.backward (engine thread), all %d steps ----' % (nm, a.steps))
            pstats.Stats(p_).sort_stats('tottime').print_stats(28)
        for key in ('tottime', 'cumtime'):
            st = pstats.Stats(pr)
            st.sort_stats(key).print_stats(45)
        return
    # ---- headline: GPU-bound number = hipGraph replays of the captured step (N = 1); eager launches are timed next to it.  With more than one
    # rank the timed path is the eager one (RCCL collectives inside a capture have never run on this pool: an exception could be caught, a hang not).
    use_graph = a.launch == 'graph' or (a.launch == 'auto' and world == 1)
    launch, graph_note = 'eager', None
    dt = dt_graph = None
    if use_graph:
        try:
            dt_graph, last_graph = timed_graph(a.epoch, a.steps)
        except Exception as e:                       # fail over to the eager path, loudly, instead of losing the bench line
            graph_note = 'graph capture failed (%s: %s); value is the eager number' % (type(e).__name__, str(e)[:200])
            torch.cuda.synchronize()
    if dt_graph is not None:                         # the capture released its memory pool: two untimed eager steps let the caching allocator re-grow before
        for _ in range(2):                           # the eager leg is timed (at B = 256 the first eager steps after a capture otherwise pay hipMalloc: 132 vs 74 ms)
            tr.train_iter(a.epoch, text, spec, target, vid)
    clock = dict(busy=0.0, steps=0)
    th.comm_clock = [] if world > 1 else None
    dt_eager, last_eager = timed_eager(a.epoch, a.steps, clock)
    comm_events, th.comm_clock = th.comm_clock, None
    # Two labelled legs over the same kernels, the same data and exactly `steps` steps between barriers + device syncs: `eager` (the default launch form:
    # what a training loop runs and what every N > 1 rank runs) and, at N = 1, `graph_replay` (the whole step captured into one hipGraph).  `value` is the
    # DEFAULT form's number -- eager -- unless --launch graph asks for the replay; it is never the minimum of the two.
    if dt_graph is not None and a.launch == 'graph':
        dt, last, launch = dt_graph, last_graph, 'hipGraph replay'
    else:
        dt, last = dt_eager, last_eager
    if dt_graph is not None and graph_note is None:
        graph_note = ('value = the %s leg; both legs run the same %d steps of the same step function between barriers + device syncs: eager launches %.3f ms/step, '
                      'hipGraph replay %.3f ms/step' % ('hipGraph replay' if a.launch == 'graph' else 'eager (default launch form)', a.steps, dt_eager / a.steps * 1e3, dt_graph / a.steps * 1e3))
    graph_leg = None if dt_graph is None else dict(ms_per_step=round(dt_graph / a.steps * 1e3, 3), value=round(a.batch * T * world / (dt_graph / a.steps), 1))
    eager = dict(ms_per_step=round(dt_eager / a.steps * 1e3, 3), value=round(a.batch * T * world / (dt_eager / a.steps), 1),
                 host_ms_per_step=round(clock['busy'] / max(clock['steps'], 1) * 1e3, 3),
                 note='host_ms_per_step = wall time inside train_iter minus the final wait for the loss read-back (python + autograd walk + launches)')
    # ---- roofline leg: a SEPARATE, untimed pass of eager steps with HIP events around the individual launches (on their launch stream)
    k_roof = 0 if a.primary_only and a.no_roofline else max(2, min(a.steps, 10))
    ops.ktimer.enabled = True
    ops.ktimer.reset()
    for _ in range(k_roof):
        tr.train_iter(a.epoch, text, spec, target, vid)
    sync()
    ops.ktimer.enabled = False
    tr.sync()
    # ---- secondary numbers, timed like the primary over the same number of steps: warm-up phase (epoch <= loss_warmup: no D update / no D-phase
    # chain), the same GAN-phase step with EVERY matrix product on the exact fp32 MFMA (ha2g_gemm_set_mode(0)), and with round 3's arithmetic (mode 6:
    # two-piece split backward = 16 operand mantissa bits, fp32-MFMA forward) -- the default (mode 70) runs every split product on three pieces
    # They are timed with EAGER launches (compare with `eager.ms_per_step`, which is within ~2 % of the replay number: the step is GPU-bound):
    # a second / third whole-step capture in one process segfaulted inside hipStreamEndCapture on the largest graphs (expressive, ~3 400 nodes,
    # ROCm 7.0.2) -- a crash there would lose the whole line, so exactly ONE capture per process is made, the headline's.
    timed = lambda ep, n: timed_eager(ep, n)
    launch2 = 'eager'
    ms_warm = ms_exact = ms_m2 = ms_m6 = float('nan')
    if not a.primary_only:
        for _ in range(2):
            tr.train_iter(0, text, spec, target, vid)
        ms_warm = timed(0, a.steps)[0] / a.steps * 1e3
        _lib.ha2g_gemm_set_mode(0)
        for _ in range(2):
            tr.train_iter(a.epoch, text, spec, target, vid)
        ms_exact = timed(a.epoch, a.steps)[0] / a.steps * 1e3
        if not a.bf16:
            _lib.ha2g_gemm_set_mode(6)                   # the round-3 default: TWO-piece split backward (16-bit operand mantissa) -- narrower than the reference's arithmetic, labelled secondary
            for _ in range(2):
                tr.train_iter(a.epoch, text, spec, target, vid)
            ms_m6 = timed(a.epoch, a.steps)[0] / a.steps * 1e3
        _lib.ha2g_gemm_set_mode(default_mode)
    tmax = torch.tensor([dt], dtype=torch.float64)
    per_rank = None
    if world > 1:
        def reduce_host(t, op):                          # a few doubles: staged through the host (works over RCCL and over the rehearsal's gloo alike)
            if rehearsal:
                dist.all_reduce(t, op=op)
                return t
            d_ = t.to(dev)
            dist.all_reduce(d_, op=op)
            return d_.cpu()
        mine = torch.zeros(world, 3, dtype=torch.float64)    # per rank: ms/step of the timed leg, exposed gradient exchange ms/step, host ms/step
        exposed = float(np.mean([s_.elapsed_time(e_) for s_, e_ in comm_events])) if comm_events else float('nan')
        mine[rank] = torch.tensor([dt / a.steps * 1e3, exposed, clock['busy'] / max(clock['steps'], 1) * 1e3], dtype=torch.float64)
        per_rank = reduce_host(mine, dist.ReduceOp.SUM).tolist()
        tmax = reduce_host(tmax, dist.ReduceOp.MAX)
    dt = float(tmax.item())
    ms = dt / a.steps * 1e3
    value = a.batch * T * world / (dt / a.steps)

    if tr.cluster_retries > a.max_cluster_retries:
        sys.stderr.write('bench.py: %d cluster-GRU recoveries (> --max-cluster-retries %d): the run did not measure the default path -- no line printed\n' % (tr.cluster_retries, a.max_cluster_retries))
        sys.exit(3)
    if rank == 0:
        # ---- roofline of the named kernel: bi-GRU layer forward (ha2g_gru_layer_fwd_cluster, H=300, layers 1..3: in = 600) and its BPTT twin ----
        H = args.hidden_size
        kt = ops.ktimer.summary()
        import hashlib
        src_sha = hashlib.sha256(open(os.path.join(ROOT, 'ha2g_amd', 'csrc', 'gru_cluster.hip'), 'rb').read()).hexdigest()

        def gru_roof(key, kernel, pmc_name, bwd, three):
            if key not in kt:
                return None
            n, mean_us, rows = kt[key][:3]                                     # rows = batch rows per launch
            flops = 2.0 * rows * T * 2 * 3 * H * H + 12.0 * rows * T * 2 * H   # recurrent matmul + gate math per launch
            if bwd:       # dy + y + reserve (4 planes) read, dg (4 planes) written, W_hh once
                bytes_ = (rows * T * 2 * H * 2 + rows * T * 2 * 4 * H * 2 + 2 * 3 * H * H) * 4.0
            else:         # gi read, y + reserve written, W_hh once
                bytes_ = (rows * T * 2 * 3 * H + rows * T * 2 * H + rows * T * 2 * 4 * H + 2 * 3 * H * H) * 4.0
            ach = flops / (mean_us * 1e-6) / 1e12
            traffic = tsrc = stale = None
            pj = os.path.join(ROOT, 'profiles', pmc_name)
            if os.path.exists(pj):                         # HBM bytes per launch from separate rocprofv3 --pmc passes (same workload)
                pm = json.load(open(pj))
                if pm.get('batch_rows') == rows:
                    traffic, tsrc = pm['hbm_bytes_per_launch'], 'profiles/%s (2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes)' % pmc_name
                    stale = pm.get('source_sha256') != src_sha     # the kernel source changed since the counters were collected
            # three-piece chains execute 6 bf16 MFMA products per fp32-equivalent product: their roofline is the dense bf16 peak / 6 (the same
            # convention as roofline_conv); the fraction of the fp32 MFMA peak -- what the fp32 form of the same kernel is priced against -- rides along
            peak = round(2500.0 / 6, 1) if three else 157.3
            return dict(kernel=kernel, bound='mfma', achieved=round(ach, 3), peak=peak,
                        unit='TFLOP/s (fp32-equivalent: 6 bf16 MFMAs per product)' if three else 'TFLOP/s', frac=round(ach / peak, 4),
                        frac_of_fp32_mfma_peak=round(ach / 157.3, 4),
                        arithmetic='three-piece bf16 split of W_hh and h, fp32 accumulate (fp32-class: all 24 mantissa bits)' if three else 'fp32 MFMA',
                        traffic=traffic, traffic_source=tsrc, traffic_stale=stale, algorithmic_bytes=bytes_,
                        traffic_over_algorithmic=(round(traffic / bytes_, 3) if traffic else None), launches=n, mean_us=round(mean_us, 1),
                        batch_rows=rows, us_per_timestep=round(mean_us / T, 2), hbm_GBps_algorithmic=round(bytes_ / (mean_us * 1e-6) / 1e9, 1),
                        hbm_frac_of_8TBps=round(bytes_ / (mean_us * 1e-6) / 8e12, 4))
        three = ops.gru_fwd3_active(H, T)                  # default mode: both recurrent chains on three bf16 pieces (gru_fwd_cluster3_kernel, gru_bwd_cluster_kernel<3>)
        roof = gru_roof('gru_layer_fwd', ('gru_fwd_cluster3_kernel (ha2g_gru_layer_fwd_cluster3, H=300)' if three else
                                          'gru_fwd_cluster_kernel (ha2g_gru_layer_fwd_cluster, H=300)'), 'r06_pmc_gru_fwd.json', False, three)
        roof_bwd = gru_roof('gru_layer_bwd', ('gru_bwd_cluster_kernel<3> (ha2g_gru_layer_bwd_cluster3, H=300)' if three else
                                              'gru_bwd_cluster_kernel (ha2g_gru_layer_bwd_cluster, H=300)'), 'r06_pmc_gru_bwd.json', True, three)
        if roof_bwd is not None and not three and bwd_pieces == 2:     # mode 6: the BPTT chain on the two-piece split (3 bf16 MFMAs per product term, fp32 accumulate)
            roof_bwd['arithmetic'] = 'split-bf16 x2 (16-bit operand mantissa; frac is still priced against the fp32 MFMA peak)'
        roof_gemm = None
        if 'gemm_gi' in kt:                        # dominant dense-GEMM shape: the GRU input projections (rows x 600) . (600 x 900)^T, fp32 MFMA
            n, mean_us, _, flops = kt['gemm_gi']
            ach = flops / (n * mean_us * 1e-6) / 1e12
            if ops.PLANE_GEMM and ops.GRU_MERGE_DIRS and bwd_pieces == 3:    # default mode: both directions in ONE product on three-piece planes (the time includes the two operand-split launches)
                roof_gemm = dict(kernel='f32_to_planes x2 + pconv_q_kernel (ha2g_gemm_planes_np_f32: GRU input projections of both directions [rows,600]x[1800,600]^T on three-piece planes)',
                                 bound='mfma', achieved=round(ach, 2), peak=round(2500.0 / 6, 1), unit='TFLOP/s (fp32-equivalent: 6 bf16 MFMAs per product)',
                                 frac=round(ach / (2500.0 / 6), 4), frac_of_fp32_mfma_peak=round(ach / 157.3, 4), launches=n, mean_us=round(mean_us, 1), traffic=None)
            else:
                roof_gemm = dict(kernel='gemm_kernel<A_KC,B_KC> (ha2g_gemm_f32: GRU input projections [rows,600]x[900,600]^T)', bound='mfma',
                                 achieved=round(ach, 2), peak=157.3, unit='TFLOP/s', frac=round(ach / 157.3, 4), launches=n, mean_us=round(mean_us, 1),
                                 traffic=None)
        roof_conv = None
        if 'conv2d_fwd' in kt:                     # the largest kernel family by time: implicit-GEMM convolutions of the audio tower (forward, fp32 MFMA)
            n, mean_us, _, flops = kt['conv2d_fwd']
            ach = flops / (n * mean_us * 1e-6) / 1e12
            if bwd_pieces == 3:      # default mode: what ha2g_conv2d_fwd_f32 still serves is the 32-channel layer on the three-piece anti-phase direct kernel + the taps (fp32 implicit GEMM)
                roof_conv = dict(kernel='conv3x3_c32pp_kernel (three-piece direct kernel: the six 32-channel forward convolutions of layer 1) + gemm_kernel<A_IM,B_KC> for the tap convolutions (ha2g_conv2d_fwd_f32)',
                                 bound='mfma', achieved=round(ach, 2), peak=round(2500.0 / 6, 1), unit='TFLOP/s (fp32-equivalent: 6 bf16 MFMAs per product)',
                                 frac=round(ach / (2500.0 / 6), 4), frac_of_fp32_mfma_peak=round(ach / 157.3, 4), launches=n, mean_us=round(mean_us, 1), traffic=None)
            else:
                roof_conv = dict(kernel='gemm_kernel<A_IM,B_KC> / conv3x3_c32_kernel (ha2g_conv2d_fwd_f32, SE-ResNet34 forward convolutions on the fp32 MFMA)', bound='mfma',
                                 achieved=round(ach, 2), peak=157.3, unit='TFLOP/s', frac=round(ach / 157.3, 4), launches=n,
                                 mean_us=round(mean_us, 1), traffic=None)
        # backward matrix kernels (round 3: plane-based conv data / weight gradients of trunk layers 2-4) priced against the 3-product split-bf16
        # roofline = dense bf16 MFMA peak / 3; BatchNorm passes against HBM with their algorithmic bytes (reads of dy / x per pass + the write)
        def mfma3(key, kernel):
            if key not in kt:
                return None
            n, mean_us, _, flops = kt[key]
            ach = flops / (n * mean_us * 1e-6) / 1e12
            nprod = 6 if bwd_pieces == 3 else 3                    # bf16 MFMAs per fp32-equivalent product
            return dict(kernel=kernel, bound='mfma', achieved=round(ach, 2), peak=round(2500.0 / nprod, 1), unit='TFLOP/s (fp32-equivalent: %d bf16 MFMAs per product)' % nprod,
                        frac=round(ach / (2500.0 / nprod), 4), launches=n, mean_us=round(mean_us, 1), traffic=None)
        roof_conv_fp32 = roof_conv                      # ha2g_conv2d_fwd_f32's callers: the 32-channel layer (three-piece direct kernel in the default mode) and the taps
        if 'conv2d_fwd_planes' in kt:                   # forward convolutions of trunk layers 2-4: three-piece planes (always six MFMAs per product)
            n, mean_us, _, flops = kt['conv2d_fwd_planes']
            ach = flops / (n * mean_us * 1e-6) / 1e12
            roof_conv = dict(kernel='pconv_r_kernel<MT,BN,PPX> (3x3 stride 1: patch-resident plane kernel; 1x1 / stride 2: pconv_q_kernel) forward gather (ha2g_conv2d_fwd_planes_np_f32: forward convolutions of SE-ResNet34 layers 2-4 on producer-written three-piece planes)',
                             bound='mfma', achieved=round(ach, 2), peak=round(2500.0 / 6, 1), unit='TFLOP/s (fp32-equivalent: 6 bf16 MFMAs per product)',
                             frac=round(ach / (2500.0 / 6), 4), frac_of_fp32_mfma_peak=round(ach / 157.3, 4), launches=n, mean_us=round(mean_us, 1), traffic=None)
        roof_bwd_gemm = mfma3('conv_dgrad_planes', 'pconv_r_kernel<MT,BN,PPX> (ha2g_conv2d_dgrad_planes_np_f32: 3x3 stride-1 data gradients of trunk layers 2-4, patch-resident three-piece planes)')
        roof_bwd_wgrad = mfma3('conv_wgrad_planes', 'pconv_wgrad_kernel + wide reduce (ha2g_conv2d_wgrad_planes_np_f32: 3x3 weight gradients of trunk layers 2-4)')

        def hbm(key, kernel):
            if key not in kt:
                return None
            n, mean_us, _, nbytes = kt[key]
            ach = nbytes / (n * mean_us * 1e-6) / 1e9
            return dict(kernel=kernel, bound='hbm', achieved=round(ach, 1), peak=8000.0, unit='GB/s', frac=round(ach / 8000.0, 4), launches=n,
                        mean_us=round(mean_us, 1), traffic=None, note='algorithmic bytes per pass; tensors of layers 2-4 (<= 73 MB) are served by the 256 MB Infinity Cache')
        roof_bn = hbm('bn_bwd', 'col_partial_kernel<1> + pair_final + bn_bwd_apply_kernel (ha2g_bn_bwd[_planes]_f32: BatchNorm backward, 3 launches)')
        roof_se_bn = hbm('se_bn_bwd_apply', 'pair_final + se_bn_bwd_apply_kernel (ha2g_se_bn_bwd_apply_np_f32: SE tail + bn2 backward, apply pass of the two-pass form)')
        roof_se_bn_reduce = hbm('se_bn_bwd_reduce', 'se_bn_reduce_kernel + se_bn_mlp_bwd_kernel (ha2g_se_bn_bwd_reduce_mlp_f32: reduction pass of the two-pass form + the excitation MLP backward)')
        roof_bn_stats = hbm('bn_stats', 'col_partial_kernel<0> + bn_stats_final (ha2g_bn_stats_f32: BatchNorm forward statistics)')
        if b16_storage:                              # bf16-storage mode: the same passes over 2-byte tensors, bf16 single-plane matrix kernels
            roof_bn = hbm('bn_bwd_b16', 'col_partial_kernel<1,b16> + pair_final + bn_bwd_apply_kernel<0,b16> (ha2g_bn_bwd_b16: BatchNorm backward over bf16 tensors)')
            roof_bn_stats = hbm('bn_stats_b16', 'col_partial_kernel<0,b16> + bn_stats_final (ha2g_bn_stats_b16)')

            def mfma1(key, kernel):
                if key not in kt:
                    return None
                n, mean_us, _, flops = kt[key]
                ach = flops / (n * mean_us * 1e-6) / 1e12
                return dict(kernel=kernel, bound='mfma', achieved=round(ach, 2), peak=2500.0, unit='TFLOP/s (dense bf16)', frac=round(ach / 2500.0, 4), launches=n,
                            mean_us=round(mean_us, 1), traffic=None)
            roof_conv = mfma1('conv2d_fwd_b16', 'pconv_kernel<.,.,.,.,1,1> forward gather (ha2g_conv2d_fwd_b16: trunk convolutions, bf16 in / bf16 out)')
            roof_bwd_gemm = mfma1('conv_dgrad_b16', 'pconv_kernel<.,.,.,.,1,1> (ha2g_conv2d_dgrad_b16: trunk data gradients, bf16 in / bf16 out)')
            roof_bwd_wgrad = mfma1('conv_wgrad_b16', 'pconv_wgrad_kernel<.,1> + wide reduce (ha2g_conv2d_wgrad_b16: 3x3 weight gradients of trunk layers 2-4, single planes)')
        out = dict(metric='pose-frames/sec (train step) for hierarchy.yml B=128 T=34' if T == 34 else 'pose-frames/sec (train step), T=%d performance-only window' % T, value=round(value, 1), unit='pose-frames/s',
                   n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=round(ms, 3), higher_is_better=True, scaling='weak',
                   vs_baseline=None,
                   dtype=(('bf16 (matrix operands; audio-trunk activations and activation gradients stored as bf16; fp32 accumulate, fp32 statistics, fp32 master weights and optimizer)'
                           if b16_storage else 'bf16 (operands; fp32 accumulate, fp32 storage and master weights)') if a.bf16 else
                          {3: 'f32 (storage, accumulation; matrix products: bf16x3 split = all 24 operand mantissa bits, six bf16 MFMAs per product, fp32 accumulate (trunk convolutions forward + backward, dense products >= 4 GFLOP, both GRU chains, every backward GEMM) or the fp32 MFMA (small forward GEMMs, tap convolutions, stem))',
                           2: 'f32 storage / accumulation, forward products fp32 MFMA; backward products: bf16x2 split = 16-bit operand mantissa (NOT fp32-class), fp32 accumulate',
                           0: 'f32 (every product on the fp32 MFMA)'}[bwd_pieces]),
                   data='synthetic', rehearsal=rehearsal, launch=launch, launch_note=graph_note, rccl_world=rccl_world, eager=eager, graph_replay=graph_leg,
                   cluster_retries=tr.cluster_retries,
                   per_rank=(None if per_rank is None else dict(
                       ms_per_step=[round(r[0], 3) for r in per_rank], ms_per_step_min=round(min(r[0] for r in per_rank), 3), ms_per_step_max=round(max(r[0] for r in per_rank), 3),
                       exposed_comm_ms=[round(r[1], 3) for r in per_rank], exposed_comm_ms_max=round(max(r[1] for r in per_rank), 3),
                       host_ms_per_step=[round(r[2], 3) for r in per_rank],
                       note='exposed_comm_ms = HIP-event time from behind the last backward kernel (audio tower) to the first optimizer kernel: the gradient exchange NOT hidden '
                            'under the backward (the audio bucket by design + what is left of the generators\' / text encoder\'s asynchronous buckets + the error-word MAX-reduce)')),
                   matrix_core=('bf16 operands (1 MFMA per product), fp32 accumulate, fp32 storage and master weights; GRU recurrences fp32' if a.bf16 else
                                {3: '3-piece split-bf16 (x = p0+p1+p2 holds all 24 mantissa bits; six bf16 MFMAs per product, smallest first, fp32 accumulate: as accurate as the fp32 MFMA chain, tests/test_gpu_np3.py): forward + backward convolutions of trunk layers 1-4 (v_mfma_f32_16x16x32_bf16 / 32x32x16), dense products >= 4 GFLOP, every backward GEMM, GRU forward and BPTT chains; fp32 MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4): small forward GEMMs, tap convolutions, stem',
                                 2: 'forward: fp32 MFMA; backward GEMMs/convs/BPTT: 2-piece split-bf16 (hi+lo, 3 bf16 MFMAs per product; 4e-6 rms-rel per GEMM vs 4e-7 for fp32 MFMA)',
                                 0: 'fp32 MFMA everywhere'}[bwd_pieces]),
                   two_piece_backward=dict(ms_per_step=round(ms_m6, 3), value=round(a.batch * T * world / (ms_m6 * 1e-3), 1) if ms_m6 == ms_m6 else None, steps=a.steps, launch=launch2,
                                           arithmetic='ha2g_gemm_set_mode(6), the round-3 default: backward products on TWO bf16 pieces = 16-bit operand mantissa -- narrower than the reference\'s fp32 backward; a labelled secondary number, never `value`'),
                   exact_fp32_matrix_core=dict(ms_per_step=round(ms_exact, 3), value=round(a.batch * T * world / (ms_exact * 1e-3), 1), steps=a.steps, launch=launch2,
                                               arithmetic='every matrix product on v_mfma_f32_32x32x2_f32 / 16x16x4_f32 (exact fp32, the reference\'s arithmetic class)'),
                   warmup_phase=dict(ms_per_step=round(ms_warm, 3), value=round(a.batch * T * world / (ms_warm * 1e-3), 1), steps=a.steps, launch=launch2),
                   config=dict(workload='%s hierarchy train step, B=%d per GPU, T=%d%s, %d-d pose, '
                                        'spec (128,%d), n_words=%d, n_spk=%d, dropout 0.3, %s' % (
                                            'config_expressive/hierarchy.yml TED-Expressive' if a.expressive else 'config/hierarchy.yml TED-Gesture', a.batch, T,
                                            '' if T == 34 else ' (PERFORMANCE-ONLY window: BASELINE config 5 asks for T = 64, which the reference modules cannot produce -- SURVEY M5; T = 62 is the legal neighbour, discriminator out2 resized to Linear(%d, 1); parity holds at T = 34 only)' % (T - 6),
                                            P, spec_w, a.n_words, a.n_spk, 'GAN phase (epoch %d > loss_warmup)' % a.epoch
                                            if a.epoch > args.loss_warmup else 'warm-up phase (epoch %d)' % a.epoch),
                               global_batch=a.batch * world, parallelism='dp%d' % world,
                               word_embedding_updates='row-wise (compact gradients, lazy Adam; bit-identical to dense)' if tr.sparse_embeddings else 'dense'),
                   gru_cluster_handoff_timeouts=ops.gru_cluster_error(dev), roofline=roof, roofline_bwd=roof_bwd, roofline_gemm=roof_gemm, roofline_conv=roof_conv, roofline_conv_fp32=roof_conv_fp32, roofline_bwd_gemm=roof_bwd_gemm, roofline_bwd_wgrad=roof_bwd_wgrad, roofline_bn=roof_bn, roofline_bn_stats=roof_bn_stats, roofline_se_bn=roof_se_bn, roofline_se_bn_reduce=roof_se_bn_reduce, roofline_pass='separate untimed pass of %d eager steps, HIP events around each launch on its launch stream' % k_roof, kernel_times_us={k: [v[0], round(v[1], 1)] for k, v in kt.items()}, last_step=last)
        if world == 1 and not a.no_cpu_baseline and not a.expressive:
            out['cpu_baseline'] = cpu_baseline(a.epoch, a.n_words, a.n_spk)
        def _clean(o):                                   # NaN (a leg that was not run) is not JSON: null instead
            if isinstance(o, float) and o != o:
                return None
            if isinstance(o, dict):
                return {k: _clean(v) for k, v in o.items()}
            if isinstance(o, (list, tuple)):
                return [_clean(v) for v in o]
            return o
        print(json.dumps(_clean(out)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

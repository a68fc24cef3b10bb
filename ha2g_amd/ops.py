"""Python face of the HIP kernels: thin functional wrappers (raw pointers + current stream through the C-ABI)
and the torch.autograd.Function classes that make `loss.backward()` at the reference's call sites run the
hand-written backward kernels.  torch is used for device memory, streams and autograd bookkeeping only;
there is no fallback path -- every op here ends in a libha2g_hip.so call.
"""
import os

import torch

from ._lib import check, lib

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_SIGMOID = 0, 1, 2, 3
(OP_ADD, OP_MUL, OP_ADD_RELU, OP_RELU_BWD, OP_LEAKY_BWD, OP_SIGMOID_BWD, OP_ELU, OP_ELU_BWD, OP_REPARAM,
 OP_REPARAM_BWD_LOGVAR, OP_AXPBY, OP_LEAKY, OP_RELU, OP_SCALE, OP_MUL_SCALAR, OP_SIGMOID, OP_SIGMOID_BWD_PRE, OP_RSQRT_EPS, OP_LEAKY_A) = range(19)

_WS_BYTES = 320 << 20      # per (device, stream) scratch: sized for the largest reduction of the path (contrastive pair block + split-K partials at N = 256 x 62 rows)
_ws = {}
_tickets = {}


# The raw handle of the current stream straight from the C layer: `torch.cuda.current_stream().cuda_stream` builds a Stream object through four python
# frames per call -- 680 calls per forward and as many per backward, 8 ms of a 35 ms host step (bench.py --host-profile, profiles/r04_host_profile.txt).
_raw_stream = torch._C._cuda_getCurrentRawStream
_cur_device = torch._C._cuda_getDevice


def _stream():
    return _raw_stream(_cur_device())


_stream_objs = {}
_NO_STREAM_CACHE = False      # True: build a torch Stream object per call (the round-3 host path; A/B from python)


def cur_stream(device=None):
    """torch.cuda.current_stream(device) without building a new Stream object per call: the objects are cached by raw handle (a step makes ~900 such
    lookups for wait_stream / record_stream bookkeeping)."""
    if _NO_STREAM_CACHE:
        return torch.cuda.current_stream(device)
    idx = device.index if (device is not None and device.index is not None) else _cur_device()
    raw = _raw_stream(idx)
    so = _stream_objs.get((idx, raw))
    if so is None:
        so = _stream_objs[(idx, raw)] = torch.cuda.current_stream(idx)
    return so


def workspace(device):
    """One persistent scratch buffer per (device, stream) (split-K partials, reduction partials).  Kernels of one stream
    run in order, so they can share it; the weight-gradient side stream gets its own.  Allocated once, outside capture."""
    raw = _raw_stream(device.index if device.index is not None else _cur_device())
    key = (device.type, device.index, raw)
    if key not in _ws:
        _ws[key] = torch.empty(_WS_BYTES // 4, dtype=torch.float32, device=device)
        if device.type == 'cuda':
            # the stream's split-K arrival tickets (include/ha2g_hip.h, ABI 5): zeroed once, registered with the library for this stream -- its
            # split-K launches then add their slabs in the kernel (last arriver per tile) instead of a second reduce launch
            tk = torch.zeros(lib.ha2g_splitk_ticket_words(), dtype=torch.int32, device=device)
            with torch.cuda.device(device):
                check(lib.ha2g_splitk_set_tickets(tk.data_ptr(), tk.numel(), raw))
            _tickets[key] = tk
    return _ws[key]


class SideStream:
    """Fork/join helper: weight / bias gradients are off the critical path of back-propagation (only the optimizer needs
    them), so backward functions enqueue them on a second HIP stream where they overlap with the data-gradient chain and
    fill CUs that latency-bound kernels (GRU clusters, small GEMMs) leave idle.
        with side.section(x.device):   # side stream waits for everything enqueued so far on the main stream
            dw = gemm(...)
        ...
        side.join(x.device)            # main stream waits for the side stream (before the results are handed to autograd)
    Tensors touched by side-stream kernels must stay referenced until join()."""

    def __init__(self):
        self.enabled = os.environ.get('HA2G_SIDE_STREAM', '1') != '0'      # 0: every weight gradient in line on the main stream (A/B)
        self._streams = {}
        self._deferred = {}
        self._targets = {}                # device -> data_ptrs of the .grad buffers deferred side-stream work is still accumulating into (None in the set: unknown)
        self.allow_defer = False          # set by a caller that flushes before it touches the gradient buffers (train_hierarchy._train_iter)

    def stream(self, device):
        key = (device.type, device.index)
        if key not in self._streams:
            self._streams[key] = torch.cuda.Stream(device)
        return self._streams[key]

    def section(self, device):
        import contextlib
        if not self.enabled:
            return contextlib.nullcontext()
        s = self.stream(device)
        s.wait_stream(cur_stream(device))
        return torch.cuda.stream(s)

    def join(self, device):
        if self.enabled:
            cur_stream(device).wait_stream(self.stream(device))
            self._deferred.pop((device.type, device.index), None)
            self._targets.pop((device.type, device.index), None)

    def defer(self, device, keep, targets=None):
        """Instead of join(): the side-stream work enqueued so far only ACCUMULATES into installed .grad buffers (nothing the main stream reads before the
        optimizer / the gradient exchange), so the main stream does not wait here; `keep` stays referenced until flush() / the next join().  The
        caller of backward() owes a flush() before anything on the main stream touches those buffers: only a caller that sets `allow_defer`
        (train_hierarchy._train_iter) gets this -- a plain `loss.backward(); optimizer.step()` at the reference's call sites keeps the joins."""
        if self.enabled:
            key = (device.type, device.index)
            self._deferred.setdefault(key, []).append(keep)
            ptrs = self._targets.setdefault(key, set())
            if targets is None:
                ptrs.add(None)                                   # a site that does not name its targets: any main-stream touch flushes
            else:
                ptrs.update(t.data_ptr() for t in targets if t is not None)

    def touch(self, device, *tensors):
        """Call before MAIN-stream work reads or accumulates into gradient buffers (a small Linear's in-place dW +=, an embedding backward, a gradient
        handed back to autograd's AccumulateGrad): if deferred side-stream work is still accumulating into one of them, the main stream joins first --
        two unordered read-modify-writes of one buffer would race silently."""
        ptrs = self._targets.get((device.type, device.index))
        if self.enabled and ptrs and (None in ptrs or any(t is not None and t.data_ptr() in ptrs for t in tensors)):
            self.join(device)

    def flush(self, device):
        if self.enabled and self._deferred.get((device.type, device.index)):
            self.join(device)


side = SideStream()


ACT_TAP = [None]      # diagnostics: a list here receives (kind, y > 0) for every ReLU / LeakyReLU applied by an autograd Function below, in call order
                      # (tests/test_gpu_linearised.py linearises the float64 oracle at exactly this activation pattern)


def _tap_act(kind, y, act):
    if ACT_TAP[0] is not None and act in (ACT_RELU, ACT_LEAKY):
        ACT_TAP[0].append((kind, (y > 0)))


def _count_grad_use(w):
    """Forward-time bookkeeping for deferred joins: how many gradient-carrying uses a NON-LEAF weight (a weight-normalised convolution weight) has.
    With exactly one, the dW a backward function returns goes to one consumer only -- the weight-norm backward, which runs on the side stream too --
    and autograd has nothing to add on the main stream."""
    if torch.is_grad_enabled() and w is not None and w.requires_grad and not w.is_leaf:
        base = w._base if w._base is not None else w
        base._ha2g_grad_uses = getattr(base, '_ha2g_grad_uses', 0) + 1


def _single_use_nonleaf(w):
    if w is None or w.is_leaf or not (DEFER_JOIN and side.allow_defer):
        return False
    base = w._base if w._base is not None else w
    return getattr(base, '_ha2g_grad_uses', 0) == 1


def _join_or_defer(device, in_place, keep, targets=None):
    """End of a backward function's side-stream section: when every weight / bias gradient went straight into an installed .grad buffer (nothing is
    handed to autograd, whose accumulation would run on the main stream) the main stream does not wait (SideStream.defer), else it joins.
    targets: the .grad buffers the section accumulates into (SideStream.touch orders later main-stream accumulations into the same buffers)."""
    if DEFER_JOIN and side.allow_defer and in_place:
        side.defer(device, keep, targets)
    else:
        side.join(device)


class KernelTimer:
    """HIP-event timing of selected kernel launches on the stream they are launched on (bench.py's roofline leg)."""

    def __init__(self):
        self.enabled = False
        self.recs = {}

    def reset(self):
        self.recs = {}

    def launch(self, name, fn, meta=0):
        if not self.enabled:
            return fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = fn()
        e.record()
        self.recs.setdefault(name, []).append((s, e, meta))
        return r

    def summary(self):
        """{name: (launches, mean microseconds, meta of the last launch, sum of metas)}; call after a device synchronize."""
        out = {}
        for k, v in self.recs.items():
            us = [s.elapsed_time(e) * 1e3 for s, e, _ in v]
            out[k] = (len(us), sum(us) / len(us), v[-1][2], sum(m for _, _, m in v))
        return out


ktimer = KernelTimer()


def _p(t):
    return 0 if t is None else t.data_ptr()


def _f32c(t):
    assert t.dtype == torch.float32 and t.is_cuda and t.is_contiguous(), (t.dtype, t.device, t.stride())
    return t


def _chk2d(t):
    assert t.dim() == 2 and t.stride(1) == 1 and t.dtype == torch.float32 and t.is_cuda, (t.shape, t.stride(), t.dtype)


def empty(*shape, like):
    return torch.empty(*shape, dtype=torch.float32, device=like.device)


# ------------------------------------------------------------------------------------------------
# raw wrappers
# ------------------------------------------------------------------------------------------------

def gemm(a, b, transa=False, transb=False, out=None, alpha=1.0, beta=0.0, bias=None, act=ACT_NONE, colsum_out=None, colsum_beta=0.0):
    """out[M,N] = act(alpha * op(a) @ op(b) + beta*out + bias).  a, b, out: 2-D fp32 CUDA tensors with unit
    inner stride (row stride free, so column slices of wider buffers work).
    colsum_out [M] (weight-gradient shape only: transa, not transb): also colsum_out = colsum_beta*colsum_out + a.sum(0), the layer's bias
    gradient, from the same launch."""
    _chk2d(a)
    _chk2d(b)
    M, K = (a.shape[1], a.shape[0]) if transa else a.shape
    Kb, N = (b.shape[1], b.shape[0]) if transb else b.shape
    assert K == Kb, (a.shape, b.shape, transa, transb)
    if out is None:
        assert beta == 0.0
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    _chk2d(out)
    assert out.shape == (M, N)
    ws = workspace(a.device)
    if _plane_gemm_ok(a, b, M, N, K, transa, transb, alpha, act, out):
        return _gemm_planes(a, b, M, N, K, transa, transb, out, beta, bias, act, colsum_out, colsum_beta, ws)
    if colsum_out is not None:
        assert transa and not transb and alpha == 1.0 and bias is None and act == ACT_NONE
        assert colsum_out.shape == (M,) and colsum_out.is_contiguous() and colsum_out.dtype == torch.float32
        check(lib.ha2g_gemm_wgrad_bias_f32(M, N, K, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), beta, out.data_ptr(), out.stride(0),
                                           colsum_beta, colsum_out.data_ptr(), ws.data_ptr(), ws.numel() * 4, _stream()))
        return out
    check(lib.ha2g_gemm_f32(int(transa), int(transb), M, N, K, alpha, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0),
                            beta, out.data_ptr(), out.stride(0), _p(bias), act, ws.data_ptr(), ws.numel() * 4, _stream()))
    return out


PLANE_GEMM = True      # large dense products on three-piece planes (csrc/conv_planes.hip, pconv_q_kernel)
# >= 4 GFLOP: the GRU input projections (both directions merged) and their backward; smaller products (generator head, TCN) lose to the operand
# split passes (A/B on the step: 43.24 ms vs 43.58 with 0.5 GFLOP, 44.10 without the plane GEMM; profiles/r04_ab_gemm.txt)
PLANE_GEMM_MIN_FLOP = 4e9
DEFER_JOIN = True       # BiGRU backward: no main-stream wait for in-place weight gradients (SideStream.defer / flush)


def _plane_gemm_ok(a, b, M, N, K, transa, transb, alpha, act, out):
    """Route this product to the plane GEMM?  Only in the fp32-class default mode (three pieces = the accuracy of the fp32 MFMA GEMM), only
    for products large enough to fill the chip (GRU input projections and their backward, the generator head), 16-byte aligned operands."""
    if 2.0 * M * N * K < PLANE_GEMM_MIN_FLOP or min(M, N) < 128 or K < 64:      # first: almost every product of a step is below the threshold (host time)
        return False
    if not (PLANE_GEMM and a.is_cuda and alpha == 1.0 and act in (ACT_NONE, ACT_RELU, ACT_LEAKY) and lib.ha2g_gemm_bwd_pieces() == 3):
        return False
    return all(t.data_ptr() % 16 == 0 and t.stride(0) % 4 == 0 for t in (a, b))


def _planes_2d(x, transpose):
    """x [rows][cols] fp32 -> three-piece planes for the plane GEMM: [3][rows][Kp] (Kp = cols rounded up to 32, zero padded), or with transpose
    [3][cols][Kp'] holding x^T (Kp' = rows rounded up to 32)"""
    rows, cols = x.shape
    R, Kv = (cols, rows) if transpose else (rows, cols)
    Kp = (Kv + 31) // 32 * 32
    pl = torch.empty(3, R, Kp, dtype=torch.bfloat16, device=x.device)
    check(lib.ha2g_f32_to_planes_2d_np(x.data_ptr(), x.stride(0), rows, cols, pl.data_ptr(), pl.stride(0), Kp, 3, int(transpose), _stream()))
    return pl


def _gemm_planes(a, b, M, N, K, transa, transb, out, beta, bias, act, colsum_out, colsum_beta, ws):
    ap = _planes_2d(a, transa)                             # [3][M][Kp]: a is [M][K], or stored [K][M]
    bp = _planes_2d(b, not transb)                         # [3][N][Kp]: b is stored [N][K] (transb), or [K][N]
    ktimer.launch('gemm_planes', lambda: check(lib.ha2g_gemm_planes_np_f32(
        ap.data_ptr(), ap.stride(0), ap.shape[2], bp.data_ptr(), bp.stride(0), bp.shape[2], 3, M, N, K, beta, out.data_ptr(), out.stride(0),
        _p(bias), act, ws.data_ptr(), ws.numel() * 4, _stream())), 2.0 * M * N * K)
    if colsum_out is not None:                             # the layer's bias gradient (rode on the weight-gradient GEMM's dY tiles before)
        colsum(a, out=colsum_out, beta=colsum_beta)
    return out


def _gemm_planes_pre(a, bp, M, N, K, out, beta=0.0, bias=None, act=ACT_NONE):
    """_gemm_planes with the B operand ALREADY split: bp [3][N][Kp] piece planes of the [N][K] operand (K contiguous, zero padded to Kp = K rounded up to
    32) -- built for a whole GRU stack by one split2d_multi launch.  a [M][K] fp32 is split here."""
    ap = _planes_2d(a, False)
    assert ap.shape[2] == bp.shape[2] and bp.shape[1] == N
    ws = workspace(a.device)
    ktimer.launch('gemm_planes', lambda: check(lib.ha2g_gemm_planes_np_f32(
        ap.data_ptr(), ap.stride(0), ap.shape[2], bp.data_ptr(), bp.stride(0), bp.shape[2], 3, M, N, K, beta, out.data_ptr(), out.stride(0),
        _p(bias), act, ws.data_ptr(), ws.numel() * 4, _stream())), 2.0 * M * N * K)
    return out


def split2d_multi(mats, windows, ps, ldp, wcols, transpose):
    """ha2g_f32_to_planes_2d_multi_np: mats = 2-D fp32 tensors (unit inner stride), windows = data pointers (ints) of their windows inside ONE plane set
    (piece stride ps, row stride ldp elements), wcols = window widths"""
    import numpy as np
    n = len(mats)
    xp = np.array([m.data_ptr() for m in mats], np.int64)
    ld = np.array([m.stride(0) for m in mats], np.int64)
    rows = np.array([m.shape[0] for m in mats], np.int32)
    cols = np.array([m.shape[1] for m in mats], np.int32)
    wp = np.array(windows, np.int64)
    wc = np.array(wcols, np.int32)
    check(lib.ha2g_f32_to_planes_2d_multi_np(xp.ctypes.data, ld.ctypes.data, rows.ctypes.data, cols.ctypes.data, wp.ctypes.data, wc.ctypes.data, ps, ldp,
                                             n, 3, int(transpose), _stream()))


def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def gemm_grouped(a, b, transa=False, transb=False, out=None, alpha=1.0, beta=0.0, bias=None, act=ACT_NONE, colsum_out=None, colsum_beta=0.0):
    """G independent GEMMs of one shape in one launch (ha2g_gemm_grouped_f32).  a, b, out, bias, colsum_out: lists of G tensors, or ONE
    stacked tensor with a leading group dimension ([G, rows, cols] operands / outputs, [G, N] bias, [G, M] column sums); per group the
    semantics of gemm()."""
    def parts(t, nd):
        if t is None or isinstance(t, (list, tuple)):
            return t
        assert t.dim() == nd + 1
        return [t[g] for g in range(t.shape[0])]
    A, Bm = parts(a, 2), parts(b, 2)
    G = len(A)
    assert len(Bm) == G and 1 <= G <= 8
    a0, b0 = A[0], Bm[0]
    _chk2d(a0); _chk2d(b0)
    sha, shb, lda, ldb = a0.shape, b0.shape, a0.stride(0), b0.stride(0)
    for i in range(1, G):                                  # one pass (this wrapper runs 60+ times per step: host time)
        ta, tb = A[i], Bm[i]
        assert ta.shape == sha and ta.stride() == (lda, 1) and tb.shape == shb and tb.stride() == (ldb, 1) and ta.dtype == tb.dtype == torch.float32, \
            'gemm_grouped: every group must have the shape, row stride and dtype of group 0'
    M, K = (sha[1], sha[0]) if transa else sha
    Kb, N = (shb[1], shb[0]) if transb else shb
    assert K == Kb
    ret = out
    if out is None:
        assert beta == 0.0
        ret = out = torch.empty(G, M, N, dtype=torch.float32, device=a0.device)
    C = parts(out, 2)
    ldc = C[0].stride(0)
    assert len(C) == G
    for t in C:
        assert t.shape == (M, N) and t.stride() == (ldc, 1)
    bias_l = parts(bias, 1)
    cs_l = parts(colsum_out, 1)
    if cs_l is not None:
        assert transa and not transb and all(t is None or (t.shape == (M,) and t.is_contiguous()) for t in cs_l)
    ws = workspace(A[0].device)
    keep = (_ptr_array(A), _ptr_array(Bm), _ptr_array(C), _ptr_array(bias_l) if bias_l is not None else None,
            _ptr_array(cs_l) if cs_l is not None else None)
    check(lib.ha2g_gemm_grouped_f32(G, int(transa), int(transb), M, N, K, alpha, keep[0], lda, keep[1], ldb, beta,
                                    keep[2], ldc, keep[3], act, keep[4], colsum_beta, ws.data_ptr(), ws.numel() * 4, _stream()))
    return ret


def gemm_grouped_raw(dev, G, transa, transb, M, N, K, pa, lda, pb, ldb, pc, ldc, beta=0.0, pcs=None, cs_beta=0.0, alpha=1.0, pbias=None, act=ACT_NONE):
    """gemm_grouped on raw device addresses (lists of G ints: fp32 operands with unit inner stride, row strides lda / ldb / ldc in elements) -- for callers
    that address column / row blocks of buffers they own (BiGRUFunction.backward: 22 slice views per layer were host time on the host-bound phase of the
    step).  The CALLER keeps the buffers alive and vouches for shapes; nothing is checked here beyond the group count."""
    import ctypes
    assert 1 <= G <= 8 and len(pa) == len(pb) == len(pc) == G
    arr = ctypes.c_void_p * G
    ws = workspace(dev)
    keep = (arr(*pa), arr(*pb), arr(*pc), arr(*pbias) if pbias is not None else None, arr(*pcs) if pcs is not None else None)
    check(lib.ha2g_gemm_grouped_f32(G, int(transa), int(transb), M, N, K, alpha, keep[0], lda, keep[1], ldb, beta, keep[2], ldc, keep[3], act, keep[4], cs_beta,
                                    ws.data_ptr(), ws.numel() * 4, _stream()))


def colsum(x, out=None, beta=0.0):
    _chk2d(x)
    if out is None:
        out = torch.empty(x.shape[1], dtype=torch.float32, device=x.device)
    check(lib.ha2g_colsum_f32(x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], out.data_ptr(), beta,
                              workspace(x.device).data_ptr(), _stream()))
    return out


def eltwise(op, a, b=None, c=None, out=None, alpha=1.0, beta=0.0):
    a = _f32c(a)
    if out is None:
        out = torch.empty_like(a)
    check(lib.ha2g_eltwise_f32(op, a.data_ptr(), _p(b), _p(c), out.data_ptr(), a.numel(), alpha, beta, _stream()))
    return out


def act_bwd(dy, y, act):
    dy = dy.contiguous()
    if act == ACT_NONE:
        return dy
    op = {ACT_RELU: OP_RELU_BWD, ACT_LEAKY: OP_LEAKY_BWD, ACT_SIGMOID: OP_SIGMOID_BWD}[act]
    return eltwise(op, dy, y)


# ------------------------------------------------------------------------------------------------
# RNG (dropout) -- device-resident Philox state so a captured graph draws new masks on each replay
# ------------------------------------------------------------------------------------------------

class Rng:
    def __init__(self):
        self.state = None
        self.call = 0
        self.step_token = 0          # host-side count of end_step() / seed() calls: dropout backward checks it re-draws in the same step

    def seed(self, device, seed):
        self.state = torch.tensor([seed, 0], dtype=torch.int64, device=device)
        self.call = 0
        self.step_token += 1

    def begin_step(self):
        self.call = 0

    def end_step(self):
        if self.state is not None:
            check(lib.ha2g_rng_advance(self.state.data_ptr(), _stream()))
            self.step_token += 1

    def next_id(self):
        self.call += 1
        return self.call


rng = Rng()


DROP_TAP = [None]     # diagnostics: a list here receives (call id, p, shape) of every dropout APPLIED in a forward (tests/test_gpu_linearised.py re-draws the masks
                      # of a step from a copy of rng.state and hands them to the float64 oracle: the step with dropout ON against the oracle with the same masks)


def _tap_drop(call, p, shape):
    if DROP_TAP[0] is not None:
        DROP_TAP[0].append((int(call), float(p), tuple(int(d) for d in shape)))


def redraw_mask(state, call, p, shape):
    """the pre-scaled keep mask ha2g_dropout_f32 draws for (state = a uint64[2] {seed, step} device tensor, call id) over a tensor of `shape`"""
    mask = torch.empty(shape, dtype=torch.float32, device=state.device)
    check(lib.ha2g_dropout_f32(0, 0, mask.data_ptr(), mask.numel(), p, state.data_ptr(), call, _stream()))
    return mask


class DropoutFunction(torch.autograd.Function):
    """out = x * mask, mask = keep / (1 - p) drawn from the device-resident Philox state keyed by (seed, step, call id, element).  The mask is
    NOT stored: the backward re-draws it by running the same kernel on dy with the same call id (the state only advances at the end of the
    step) -- one tensor less written in the forward and read in the backward.  A backward issued after the step counter moved would silently
    draw a different mask, so that is an error."""

    @staticmethod
    def forward(ctx, x, p):
        x = _f32c(x.contiguous())
        if rng.state is None or rng.state.device != x.device:
            rng.seed(x.device, 0x5EED)
        out = torch.empty_like(x)
        ctx.p, ctx.call, ctx.step_token = p, rng.next_id(), rng.step_token
        check(lib.ha2g_dropout_f32(x.data_ptr(), out.data_ptr(), None, x.numel(), p, rng.state.data_ptr(), ctx.call, _stream()))
        _tap_drop(ctx.call, p, x.shape)
        return out

    @staticmethod
    def backward(ctx, dy):
        if ctx.step_token != rng.step_token:
            raise RuntimeError('ha2g_amd dropout: backward after the RNG step advanced (rng.end_step()): the mask cannot be re-drawn')
        dy = _f32c(dy.contiguous())
        dx = torch.empty_like(dy)
        check(lib.ha2g_dropout_f32(dy.data_ptr(), dx.data_ptr(), None, dy.numel(), ctx.p, rng.state.data_ptr(), ctx.call, _stream()))
        return dx, None


def dropout(x, p, training):
    if not training or p <= 0.0:
        return x
    return DropoutFunction.apply(x, p)


class DropSpec:
    """A dropout attached to the op in front of / behind it (round 6: the TCN blocks' dropouts, model/tcn.py:21-31, fused with their neighbours).
    conv1d_tm / grouped_linear(..., act=ACT_RELU, drop=spec): y = dropout(relu(conv)) -- the forward applies the mask unless spec.deferred (then the CONSUMER
    does: add_relu(y, res, drop=spec) = relu(mask * y + res) in one launch, and hands back the gradient w.r.t. the masked value); the backward applies
    dropout' and ReLU' in ONE launch (ha2g_dropout_fused_f32 mode 2) instead of two.  The mask is the one ops.dropout draws for the same call id."""
    __slots__ = ('p', 'call', 'token', 'deferred')

    def __init__(self, p, call, token, deferred):
        self.p, self.call, self.token, self.deferred = p, call, token, deferred


FUSE_TCN_DROPOUT = True
FUSE_IM2COL_DROPOUT = True      # a deferred DropSpec handed to the NEXT convolution (conv1d_tm / grouped_conv1d_tm in_drop=spec): its im2col applies the mask


def make_drop(p, training, device, deferred=False):
    if not training or p <= 0.0 or not FUSE_TCN_DROPOUT:
        return None
    if rng.state is None or rng.state.device != device:
        rng.seed(device, 0x5EED)
    return DropSpec(p, rng.next_id(), rng.step_token, deferred)


def im2col_drop_ok(C, k):
    """can the convolution that consumes dropout(y) (y [.., C], kernel k) apply the mask in its im2col?"""
    return bool(FUSE_TCN_DROPOUT and FUSE_IM2COL_DROPOUT and lib.ha2g_im2col1d_drop_supported(C, k))


def _im2col(x, col, B, T, C, k, dil, pad_left, To, in_drop):
    if in_drop is None:
        check(lib.ha2g_im2col1d_f32(x.data_ptr(), col.data_ptr(), B, T, C, k, dil, pad_left, To, _stream()))
    else:
        if in_drop.token != rng.step_token:
            raise RuntimeError('ha2g_amd dropout: a DropSpec of an earlier RNG step')
        check(lib.ha2g_im2col1d_drop_f32(x.data_ptr(), col.data_ptr(), B, T, C, k, dil, pad_left, To, in_drop.p, rng.state.data_ptr(), in_drop.call, _stream()))
        _tap_drop(in_drop.call, in_drop.p, x.shape)


def _drop_apply(y, spec):
    out = torch.empty_like(y)
    check(lib.ha2g_dropout_f32(y.data_ptr(), out.data_ptr(), None, y.numel(), spec.p, rng.state.data_ptr(), spec.call, _stream()))
    _tap_drop(spec.call, spec.p, y.shape)
    return out


def _drop_relu_bwd(dy, y, spec):
    """d(pre-activation) = (y > 0) * mask * dy: the dropout's backward and the ReLU' of the op in front of it, one launch"""
    if spec.token != rng.step_token:
        raise RuntimeError('ha2g_amd dropout: backward after the RNG step advanced (rng.end_step()): the mask cannot be re-drawn')
    dy = _f32c(dy.contiguous())
    out = torch.empty_like(dy)
    check(lib.ha2g_dropout_fused_f32(dy.data_ptr(), y.data_ptr(), out.data_ptr(), dy.numel(), spec.p, rng.state.data_ptr(), spec.call, 2, _stream()))
    return out


DX_FIRST = False          # the same order in the TCN convolutions' and the grouped layers' backward: measured, no gain (34.16 vs 34.12 ms), off
GRU_DX_FIRST = True       # BiGRU backward: a layer's dX product is enqueued before its weight-gradient products fork to the side queue (A/B switch)
GRU_MASK_SPEC = True      # the GRU's inter-layer dropouts as specs (mask re-drawn where it is applied: one launch forward, one backward, no mask tensor)


def gru_drop_spec(p, device):
    """nn.GRU's inter-layer dropout (model/hierarchy_net.py:87) without a mask tensor: BiGRUFunction applies dropout(y) in ONE launch per layer (a mask
    launch + a multiply, 31 MB written and re-read per layer at 384 rows before) and re-draws the mask over the gradient-carrying row slice in the backward"""
    if rng.state is None or rng.state.device != device:
        rng.seed(device, 0x5EED)
    return DropSpec(p, rng.next_id(), rng.step_token, False)


def dropout_mask(shape, p, device):
    """Pre-scaled keep mask only (used for the GRU inter-layer dropout)."""
    if rng.state is None or rng.state.device != device:
        rng.seed(device, 0x5EED)
    mask = torch.empty(shape, dtype=torch.float32, device=device)
    check(lib.ha2g_dropout_f32(0, 0, mask.data_ptr(), mask.numel(), p, rng.state.data_ptr(), rng.next_id(), _stream()))
    return mask


# ------------------------------------------------------------------------------------------------
# Linear / embedding / pointwise
# ------------------------------------------------------------------------------------------------

def _null():
    import contextlib
    return contextlib.nullcontext()


DIRECT_GRAD = True     # accumulate weight gradients straight into an existing `.grad` buffer (beta = 1 epilogues) instead of
                       # returning a temporary that autograd then adds with its own kernel (one at::add per parameter)


GROUP_GRU_WGRAD = True   # the two directions' weight gradients of a GRU layer as grouped launches
FUSE_BIAS_GRAD = True     # Linear / Conv1d: bias gradient = column sums taken by the weight-gradient GEMM (ha2g_gemm_wgrad_bias_f32)


def _grad_target(param):
    """The parameter's installed gradient buffer if a kernel may accumulate into it directly, else None."""
    g = param.grad if DIRECT_GRAD and param.is_leaf else None
    if g is None or g.shape != param.shape or g.dtype != torch.float32 or g.stride() != param.stride():
        return None
    return g


class LinearFunction(torch.autograd.Function):
    """y = act(x W^T + b); x [..., K], W [N, K] (nn.Linear layout)."""

    @staticmethod
    def forward(ctx, x, w, b, act):
        x2 = x.reshape(-1, x.shape[-1])
        if x2.stride(1) != 1:
            x2 = x2.contiguous()
        w_in = w
        w = w.contiguous()
        y = gemm(x2, w, transb=True, bias=b, act=act)
        _tap_act('linear', y, act)
        ctx.act = act
        ctx.xshape = x.shape
        ctx.has_b = b is not None
        ctx.bias_ref = b
        ctx.wref = w_in
        ctx.save_for_backward(x2, w, y if act != ACT_NONE else None)
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        dy2 = act_bwd(dy.reshape(-1, dy.shape[-1]), y, ctx.act)
        if dy2.stride(1) != 1:
            dy2 = dy2.contiguous()
        dx = dw = db = None
        big = dy2.shape[0] >= 1024                       # tiny layers: the fork/join costs more than it hides
        tgt = None
        if DX_FIRST and big and ctx.needs_input_grad[0]:
            dx = gemm(dy2, w).view(ctx.xshape)
        with (side.section(dy2.device) if big else _null()):
            want_b = ctx.has_b and ctx.needs_input_grad[2]
            tb = _grad_target(ctx.bias_ref) if (want_b and ctx.bias_ref is not None) else None
            if not big and dy2.is_cuda:
                # main-stream accumulation: ordered behind deferred side-stream accumulations into the same buffers (the same Linear applied at a big
                # row count earlier in this backward)
                side.touch(dy2.device, tb, _grad_target(ctx.wref) if ctx.needs_input_grad[1] else None)
            fuse_b = want_b and ctx.needs_input_grad[1] and FUSE_BIAS_GRAD        # bias gradient from the weight-gradient launch
            if fuse_b and tb is None:
                db = torch.empty(dy2.shape[1], dtype=torch.float32, device=dy2.device)
            cs = dict(colsum_out=tb if tb is not None else db, colsum_beta=1.0 if tb is not None else 0.0) if fuse_b else {}
            if ctx.needs_input_grad[1]:
                tgt = _grad_target(ctx.wref)
                if tgt is not None and tgt.is_contiguous():
                    gemm(dy2, x2, transa=True, out=tgt, beta=1.0, **cs)
                else:
                    dw = gemm(dy2, x2, transa=True, **cs)
            if want_b and not fuse_b:
                if tb is not None:
                    colsum(dy2, out=tb, beta=1.0)
                else:
                    db = colsum(dy2)
        if ctx.needs_input_grad[0] and dx is None:
            dx = gemm(dy2, w).view(ctx.xshape)
        if big:
            _join_or_defer(dy2.device, dw is None and db is None, (dy2, x2), (tgt, tb))
        return dx, dw, db, None


def linear(x, w, b=None, act=ACT_NONE):
    _count_grad_use(w)
    return LinearFunction.apply(x, w, b, act)


class EmbeddingFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tok, w):
        tok = tok.contiguous()
        out = torch.empty(*tok.shape, w.shape[1], dtype=torch.float32, device=w.device)
        st = getattr(w, '_ha2g_sparse', None)
        ctx.sparse = st
        if st is not None:
            # row-wise (sparse) table: compact the batch's ids, replay the zero-gradient Adam updates those rows missed since they were
            # last touched (dense Adam moves every row every step), THEN read them
            n = tok.numel()
            uniq = torch.empty(n + 1, dtype=torch.int64, device=w.device)
            remap = torch.empty(n, dtype=torch.int64, device=w.device)
            cpos = torch.empty(n, dtype=torch.int32, device=w.device)
            count = torch.empty(1, dtype=torch.int32, device=w.device)
            check(lib.ha2g_unique_tokens(tok.data_ptr(), n, st.map.data_ptr(), cpos.data_ptr(), uniq.data_ptr(), remap.data_ptr(), count.data_ptr(), _stream()))
            st.catch_up(uniq, count, n + 1)
            ctx.compact = (uniq, remap, count)
            if torch.is_grad_enabled() and w.requires_grad:
                st.prefetch_count(count)                   # data parallel: the row count leaves for the host NOW, the exchange after the backward needs it
        check(lib.ha2g_embedding_fwd_f32(tok.data_ptr(), w.data_ptr(), out.data_ptr(), tok.numel(), w.shape[1], _stream()))
        ctx.save_for_backward(tok)
        ctx.wshape = w.shape
        ctx.wref = w
        return out

    @staticmethod
    def backward(ctx, dy):
        (tok,) = ctx.saved_tensors
        dy = dy.contiguous()
        if ctx.sparse is not None:                       # compact gradient: one summed row per distinct id, handed to the optimizer
            uniq, remap, count = ctx.compact
            n, C = tok.numel(), ctx.wshape[1]
            vals = torch.zeros(n + 1, C, dtype=torch.float32, device=dy.device)
            check(lib.ha2g_embedding_bwd_f32(remap.data_ptr(), dy.data_ptr(), vals.data_ptr(), n, C, 0, workspace(dy.device).data_ptr(), _stream()))
            ctx.sparse.pending.append((uniq, count, vals))
            return None, None
        tgt = _grad_target(ctx.wref)
        direct = tgt is not None and tgt.is_contiguous()
        if direct and dy.is_cuda:
            side.touch(dy.device, tgt)
        dw = tgt if direct else torch.zeros(ctx.wshape, dtype=torch.float32, device=dy.device)
        check(lib.ha2g_embedding_bwd_f32(tok.data_ptr(), dy.data_ptr(), dw.data_ptr(), tok.numel(), ctx.wshape[1], 0,
                                         workspace(dy.device).data_ptr(), _stream()))          # the kernel accumulates (dW +=)
        return None, (None if direct else dw)


def embedding(tok, w):
    return EmbeddingFunction.apply(tok, w)


class EltUnary(torch.autograd.Function):
    """ELU (alpha = 1) -- the only standalone unary activation on the path (ResNetSE34V2.py:200-201)."""

    @staticmethod
    def forward(ctx, x):
        y = eltwise(OP_ELU, x.contiguous())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        return eltwise(OP_ELU_BWD, dy.contiguous(), ctx.saved_tensors[0])


def elu(x):
    return EltUnary.apply(x)


class AddReluFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, drop=None):
        a, b = _f32c(a.contiguous()), _f32c(b.contiguous())
        if drop is not None:                               # relu(mask * a + b): a's producer left its dropout to this launch (DropSpec.deferred)
            y = torch.empty_like(a)
            check(lib.ha2g_dropout_fused_f32(a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), drop.p, rng.state.data_ptr(), drop.call, 1, _stream()))
            _tap_drop(drop.call, drop.p, a.shape)
        else:
            y = eltwise(OP_ADD_RELU, a, b)
        _tap_act('add_relu', y, ACT_RELU)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        g = eltwise(OP_RELU_BWD, dy.contiguous(), ctx.saved_tensors[0])
        return g, g, None                                  # with a deferred dropout: the gradient w.r.t. the MASKED a -- its producer's backward applies the mask


def add_relu(a, b, drop=None):
    assert drop is None or drop.deferred
    return AddReluFunction.apply(a, b, drop)


class ReparamFunction(torch.autograd.Function):
    """z = mu + eps * exp(0.5 logvar)  (model/embedding_net.py:10-13)."""

    @staticmethod
    def forward(ctx, mu, logvar, eps):
        mu, logvar, eps = mu.contiguous(), logvar.contiguous(), eps.contiguous()
        ctx.save_for_backward(logvar, eps)
        return eltwise(OP_REPARAM, mu, logvar, eps)

    @staticmethod
    def backward(ctx, dz):
        logvar, eps = ctx.saved_tensors
        dz = dz.contiguous()
        return dz, eltwise(OP_REPARAM_BWD_LOGVAR, dz, logvar, eps), None


def reparameterize(mu, logvar, eps):
    return ReparamFunction.apply(mu, logvar, eps)


class DirSumFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y):
        y = _f32c(y.contiguous())
        H = y.shape[-1] // 2
        rows = y.numel() // (2 * H)
        out = torch.empty(*y.shape[:-1], H, dtype=torch.float32, device=y.device)
        check(lib.ha2g_dirsum_f32(y.data_ptr(), out.data_ptr(), rows, H, 0, _stream()))
        return out

    @staticmethod
    def backward(ctx, d):
        d = _f32c(d.contiguous())
        H = d.shape[-1]
        out = torch.empty(*d.shape[:-1], 2 * H, dtype=torch.float32, device=d.device)
        check(lib.ha2g_dirsum_f32(d.data_ptr(), out.data_ptr(), d.numel() // H, H, 1, _stream()))
        return out


def dirsum(y):
    return DirSumFunction.apply(y)


# ------------------------------------------------------------------------------------------------
# 1-D convolutions (TCN causal dilated k=2; discriminator valid k=3) as im2col + GEMM, weight norm
# ------------------------------------------------------------------------------------------------

class WeightNormFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, v):
        g_in, v_in = g, v
        g, v = g.contiguous(), v.contiguous()
        cout, n = v.shape[0], v[0].numel()
        w, norm = torch.empty_like(v), torch.empty(cout, dtype=torch.float32, device=v.device)
        check(lib.ha2g_weight_norm_fwd_f32(g.data_ptr(), v.data_ptr(), w.data_ptr(), norm.data_ptr(), cout, n, _stream()))
        ctx.save_for_backward(g, v, norm)
        ctx.refs = (g_in, v_in)
        return w

    @staticmethod
    def backward(ctx, dw):
        g, v, norm = ctx.saved_tensors
        tg, tv = _grad_target(ctx.refs[0]), _grad_target(ctx.refs[1])
        if tg is not None and tv is not None and tg.is_contiguous() and tv.is_contiguous():     # straight into the flat .grad buffers
            # on the SIDE stream: dw may still be in flight there (a convolution backward that did not join, _single_use_nonleaf), and the result
            # only feeds the optimizer
            with side.section(dw.device):
                dw = dw.contiguous()
                check(lib.ha2g_weight_norm_bwd_f32(dw.data_ptr(), g.data_ptr(), v.data_ptr(), norm.data_ptr(), tg.data_ptr(), tv.data_ptr(),
                                                   v.shape[0], v[0].numel(), 1.0, _stream()))
            _join_or_defer(dw.device, True, (dw, g, v, norm), (tg, tv))
            return None, None
        side.flush(dw.device)
        dw = dw.contiguous()
        dg, dv = torch.empty_like(g), torch.empty_like(v)
        check(lib.ha2g_weight_norm_bwd_f32(dw.data_ptr(), g.data_ptr(), v.data_ptr(), norm.data_ptr(), dg.data_ptr(), dv.data_ptr(),
                                           v.shape[0], v[0].numel(), 0.0, _stream()))
        return dg, dv


def weight_norm(g, v):
    return WeightNormFunction.apply(g, v)


class MultiWeightNormFunction(torch.autograd.Function):
    """n weight-norm layers of one shape (n <= 32) as one launch forward and one backward: apply(n, g_0..g_{n-1}, v_0..v_{n-1}) -> n weights."""

    @staticmethod
    def forward(ctx, n, *gv):
        gs_in, vs_in = gv[:n], gv[n:]
        gs, vs = [g.contiguous() for g in gs_in], [v.contiguous() for v in vs_in]
        cout, rl = vs[0].shape[0], vs[0][0].numel()
        assert all(v.shape == vs[0].shape for v in vs)
        ws = [torch.empty_like(v) for v in vs]
        norms = torch.empty(n, cout, dtype=torch.float32, device=vs[0].device)
        check(lib.ha2g_weight_norm_multi_fwd_f32(n, _ptr_array(gs), _ptr_array(vs), _ptr_array(ws), _ptr_array([norms[i] for i in range(n)]), cout, rl,
                                                 _stream()))
        ctx.n, ctx.refs = n, (gs_in, vs_in)
        ctx.save_for_backward(norms, *gs, *vs)
        return tuple(ws)

    @staticmethod
    def backward(ctx, *dws):
        n = ctx.n
        norms, *gv = ctx.saved_tensors
        gs, vs = gv[:n], gv[n:]
        cout, rl = vs[0].shape[0], vs[0][0].numel()
        dev = vs[0].device
        tg = [_grad_target(t) for t in ctx.refs[0]]
        tv = [_grad_target(t) for t in ctx.refs[1]]
        direct = all(t is not None and t.is_contiguous() for t in tg + tv)
        if not direct:
            side.flush(dev)                                  # the dws may still be in flight on the side stream: this branch runs on the main stream
            tg, tv = [torch.empty_like(g) for g in gs], [torch.empty_like(v) for v in vs]
        # direct: on the SIDE stream (the dws were produced there, possibly without a join: _single_use_nonleaf), in place into the .grad buffers
        with (side.section(dev) if direct else _null()):
            dws = [(d.contiguous() if d is not None else torch.zeros_like(vs[i])) for i, d in enumerate(dws)]
            check(lib.ha2g_weight_norm_multi_bwd_f32(n, _ptr_array(dws), _ptr_array(gs), _ptr_array(vs), _ptr_array([norms[i] for i in range(n)]),
                                                     _ptr_array(tg), _ptr_array(tv), cout, rl, 1.0 if direct else 0.0, _stream()))
        if direct:
            _join_or_defer(dev, True, (dws, norms, gs, vs), tg + tv)
            return (None,) * (1 + 2 * n)
        return (None,) + tuple(tg) + tuple(tv)


def weight_norm_multi(gs, vs):
    """[weight_norm(g, v) for g, v in zip(gs, vs)] for same-shape layers, 32 per launch."""
    out = []
    for a in range(0, len(gs), 32):
        g_, v_ = list(gs[a:a + 32]), list(vs[a:a + 32])
        out += list(MultiWeightNormFunction.apply(len(g_), *g_, *v_))
    return out


class Conv1dFunction(torch.autograd.Function):
    """x [B,T,Cin] (time-major rows), w [Cout,Cin,k] (nn.Conv1d layout), out [B,To,Cout].
    out[b,t] = sum_kk W[:,:,kk] x[b, t - pad_left + kk*dil]."""

    @staticmethod
    def forward(ctx, x, w, b, dil, pad_left, To, act, drop=None, in_drop=None):
        x = _f32c(x.contiguous())
        ctx.refs = (w, b)
        w = w.contiguous()
        B, T, C = x.shape
        cout, _, k = w.shape
        col = torch.empty(B * To, C * k, dtype=torch.float32, device=x.device)
        _im2col(x, col, B, T, C, k, dil, pad_left, To, in_drop)      # in_drop: x is the UNMASKED output of the op in front, whose deferred dropout is applied here
        y = gemm(col, w.view(cout, C * k), transb=True, bias=b, act=act)
        _tap_act('conv1d', y.view(B, To, cout), act)
        ctx.geom = (B, T, C, k, dil, pad_left, To, act)
        ctx.has_b = b is not None
        ctx.drop = drop
        ctx.save_for_backward(col, w, y if act != ACT_NONE else None)
        out = _drop_apply(y, drop) if (drop is not None and not drop.deferred) else y      # DropSpec: dropout(relu(conv)), see there
        return out.view(B, To, cout)

    @staticmethod
    def backward(ctx, dy):
        col, w, y = ctx.saved_tensors
        B, T, C, k, dil, pad_left, To, act = ctx.geom
        cout = w.shape[0]
        dy2 = _drop_relu_bwd(dy.reshape(B * To, cout), y, ctx.drop) if ctx.drop is not None else act_bwd(dy.reshape(B * To, cout), y, act)
        dx = dw = db = tw = None
        dcol = gemm(dy2, w.view(cout, C * k)) if (DX_FIRST and ctx.needs_input_grad[0]) else None      # DX_FIRST: see there
        with side.section(dy2.device):
            want_b = ctx.has_b and ctx.needs_input_grad[2]
            tb = _grad_target(ctx.refs[1]) if want_b else None
            fuse_b = want_b and ctx.needs_input_grad[1] and FUSE_BIAS_GRAD        # bias gradient from the weight-gradient launch
            if fuse_b and tb is None:
                db = torch.empty(cout, dtype=torch.float32, device=dy2.device)
            cs = dict(colsum_out=tb if tb is not None else db, colsum_beta=1.0 if tb is not None else 0.0) if fuse_b else {}
            if ctx.needs_input_grad[1]:
                tw = _grad_target(ctx.refs[0])
                if tw is not None and tw.is_contiguous():             # a leaf weight (discriminator convs): dW += straight into .grad
                    gemm(dy2, col, transa=True, out=tw.view(cout, C * k), beta=1.0, **cs)
                else:                                                 # weight-normalised TCN convs: dw feeds weight_norm's backward
                    dw = gemm(dy2, col, transa=True, **cs).view(cout, C, k)
            if want_b and not fuse_b:
                if tb is not None:
                    colsum(dy2, out=tb, beta=1.0)
                else:
                    db = colsum(dy2)
        if ctx.needs_input_grad[0]:
            if dcol is None:
                dcol = gemm(dy2, w.view(cout, C * k))
            dx = torch.empty(B, T, C, dtype=torch.float32, device=dy.device)
            check(lib.ha2g_col2im1d_f32(dcol.data_ptr(), dx.data_ptr(), B, T, C, k, dil, pad_left, To, _stream()))
        _join_or_defer(dy2.device, (dw is None or _single_use_nonleaf(ctx.refs[0])) and db is None, (dy2, col, dw), (tw, tb))
        return dx, dw, db, None, None, None, None, None, None


def conv1d_tm(x, w, b, dil=1, pad_left=0, To=None, act=ACT_NONE, drop=None, in_drop=None):
    """in_drop: the deferred DropSpec of the op that produced x -- this convolution's im2col applies the mask (the gradient handed back is the one w.r.t. the
    masked value, as add_relu(.., drop=spec) does)"""
    if To is None:
        To = x.shape[1] + pad_left - (w.shape[2] - 1) * dil
    assert drop is None or act == ACT_RELU
    assert in_drop is None or (in_drop.deferred and im2col_drop_ok(x.shape[2], w.shape[2]))
    _count_grad_use(w)
    return Conv1dFunction.apply(x, w, b, dil, pad_left, To, act, drop, in_drop)

# ------------------------------------------------------------------------------------------------
# grouped layers: the same layer of G networks (own weights) as one launch per GEMM (ha2g_gemm_grouped_f32)
# ------------------------------------------------------------------------------------------------

class GroupedLinearFunction(torch.autograd.Function):
    """y[g] = act(x[g] W_g^T + b_g); x [G, R, K] stacked, W_g [N, K] and b_g [N] separate tensors (G parameters of G modules)."""

    @staticmethod
    def forward(ctx, x, act, G, drop, *wb):
        ws, bs = list(wb[:G]), list(wb[G:])
        has_b = bs[0] is not None
        x = _f32c(x.contiguous())
        wc = [w.contiguous() for w in ws]
        y = gemm_grouped(x, wc, transb=True, bias=bs if has_b else None, act=act)
        _tap_act('grouped_linear', y, act)
        ctx.act, ctx.G, ctx.has_b, ctx.drop = act, G, has_b, drop
        ctx.refs = (ws, bs)
        ctx.save_for_backward(x, y if act != ACT_NONE else None, *wc)
        return _drop_apply(y, drop) if (drop is not None and not drop.deferred) else y      # DropSpec: dropout(relu(.)), see there

    @staticmethod
    def backward(ctx, dy):
        x, y, *wc = ctx.saved_tensors
        G = ctx.G
        ws, bs = ctx.refs
        dy2 = _drop_relu_bwd(dy, y, ctx.drop) if ctx.drop is not None else act_bwd(dy, y, ctx.act)
        N, K = wc[0].shape
        dx = None
        dws, dbs = [None] * G, [None] * G
        targets = []
        need_w, need_b = ctx.needs_input_grad[4], ctx.has_b and ctx.needs_input_grad[4 + G]
        if DX_FIRST and ctx.needs_input_grad[0]:
            dx = gemm_grouped(dy2, wc)
        with side.section(dy2.device):
            if need_w:
                tw = [_grad_target(w) for w in ws]
                direct_w = all(t is not None and t.is_contiguous() for t in tw)
                tb = [_grad_target(b) for b in bs] if need_b else None
                direct_b = need_b and all(t is not None for t in tb)
                if need_b and not direct_b:
                    tb = torch.empty(G, N, dtype=torch.float32, device=dy2.device)
                    dbs = [tb[g] for g in range(G)]
                if direct_w:
                    out = [t.view(N, K) for t in tw]
                else:
                    out = torch.empty(G, N, K, dtype=torch.float32, device=dy2.device)
                    dws = [out[g].view(ws[g].shape) for g in range(G)]
                gemm_grouped(dy2, x, transa=True, out=out, beta=1.0 if direct_w else 0.0,
                             colsum_out=tb if need_b else None, colsum_beta=1.0 if direct_b else 0.0)
                targets = (list(tw) if direct_w else []) + (list(tb) if direct_b else [])
            elif need_b:
                for g in range(G):
                    dbs[g] = colsum(dy2[g])
        if ctx.needs_input_grad[0] and dx is None:
            dx = gemm_grouped(dy2, wc)
        _join_or_defer(dy2.device, all(t is None or _single_use_nonleaf(w) for t, w in zip(dws, ws)) and all(t is None for t in dbs), (dy2, x, dws), targets)
        return (dx, None, None, None) + tuple(dws) + tuple(dbs)


def grouped_linear(x, ws, bs=None, act=ACT_NONE, drop=None):
    G = len(ws)
    assert drop is None or act == ACT_RELU
    for w in ws:
        _count_grad_use(w)                                 # (here, not in forward(): autograd runs forward() with gradients disabled)
    return GroupedLinearFunction.apply(x, act, G, drop, *ws, *(bs if bs is not None else [None] * G))


class Im2col1dFunction(torch.autograd.Function):
    """x [B, T, C] -> columns [B*To, C*k] of a dilated causal 1-D convolution (the GEMM operand of Conv1dFunction); backward = col2im."""

    @staticmethod
    def forward(ctx, x, k, dil, pad_left, To, in_drop=None):
        x = _f32c(x.contiguous())
        B, T, C = x.shape
        col = torch.empty(B * To, C * k, dtype=torch.float32, device=x.device)
        _im2col(x, col, B, T, C, k, dil, pad_left, To, in_drop)
        ctx.geom = (B, T, C, k, dil, pad_left, To)
        return col

    @staticmethod
    def backward(ctx, dcol):
        B, T, C, k, dil, pad_left, To = ctx.geom
        dcol = dcol.contiguous()
        dx = torch.empty(B, T, C, dtype=torch.float32, device=dcol.device)
        check(lib.ha2g_col2im1d_f32(dcol.data_ptr(), dx.data_ptr(), B, T, C, k, dil, pad_left, To, _stream()))
        return dx, None, None, None, None, None


def grouped_conv1d_tm(x, ws, bs, dil=1, pad_left=0, To=None, act=ACT_NONE, drop=None, in_drop=None):
    """x [G, B, T, C] (the inputs of G same-shape convolutions), ws: G weights [Cout, C, k], bs: G biases -> [G, B, To, Cout]."""
    G, B, T, C = x.shape
    cout, _, k = ws[0].shape
    if To is None:
        To = T + pad_left - (k - 1) * dil
    assert in_drop is None or (in_drop.deferred and im2col_drop_ok(C, k))
    col = Im2col1dFunction.apply(x.reshape(G * B, T, C), k, dil, pad_left, To, in_drop)
    y = grouped_linear(col.view(G, B * To, C * k), [w.reshape(cout, C * k) for w in ws], bs, act, drop)
    return y.view(G, B, To, cout)



# ------------------------------------------------------------------------------------------------
# BatchNorm over [rows, C]
# ------------------------------------------------------------------------------------------------

def bn_stats(x2, running_mean, running_var, momentum, eps):
    rows, C = x2.shape
    mean, invstd = empty(C, like=x2), empty(C, like=x2)
    ktimer.launch('bn_stats', lambda: check(lib.ha2g_bn_stats_f32(
        x2.data_ptr(), rows, C, mean.data_ptr(), invstd.data_ptr(), _p(running_mean), _p(running_var), momentum, eps,
        workspace(x2.device).data_ptr(), _stream())), 4.0 * rows * C)
    return mean, invstd


def bn_stats_finalize(part, nblk, rows, C, running_mean, running_var, momentum, eps):
    """mean / invstd (+ running statistics) from the per-block sums [2, C, nblk] (float64) a convolution's epilogue wrote
    (ha2g_conv2d_fwd_planes_np_stats_f32): bn_stats without its pass over the tensor."""
    mean, invstd = torch.empty(C, dtype=torch.float32, device=part.device), torch.empty(C, dtype=torch.float32, device=part.device)
    ktimer.launch('bn_stats_finalize', lambda: check(lib.ha2g_bn_stats_finalize_f32(
        part.data_ptr(), nblk, rows, C, mean.data_ptr(), invstd.data_ptr(), _p(running_mean), _p(running_var), momentum, eps, _stream())), 0)
    return mean, invstd


def bn_apply(x2, mean, invstd, gamma, beta, act=ACT_NONE, out=None):
    rows, C = x2.shape
    if out is None:
        out = torch.empty_like(x2)
    check(lib.ha2g_bn_apply_f32(x2.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(),
                                rows, C, act, _stream()))
    return out


def bn_apply_pool(x, mean, invstd, gamma, beta):
    """x [N,H,W,C] -> (bn(x), mean over H*W of bn(x) [N,C]) in one pass (BatchNorm apply + SE squeeze)."""
    N, H, W, C = x.shape
    y, pooled = torch.empty_like(x), empty(N, C, like=x)
    ws = workspace(x.device)
    assert lib.ha2g_bn_apply_pool_workspace_floats(N, H * W, C) <= ws.numel()
    check(lib.ha2g_bn_apply_pool_f32(x.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                                     N, H * W, C, pooled.data_ptr(), ws.data_ptr(), _stream()))
    return y, pooled


def bn_bwd(dy2, x2, mean, invstd, gamma, need_dx=True, relu_mask=False, acc=None, planes=False, partials=None):
    """acc = (gamma.grad, beta.grad) buffers to accumulate into directly (the returned dgamma/dbeta are the fresh sums).
    planes: dx is also written as (hi, lo) bf16 planes for the plane-based split-bf16 consumers (csrc/conv_planes.hip) and the return value
    grows to (dx, dgamma, dbeta, (hi, lo)); with need_dx=False ONLY the planes are written (dx is None).
    partials = (stat_part [2, C, nblk] float64, nblk): the tile sums of (dy, dy * xhat) the producer of dy left behind (round 6): no column pass."""
    rows, C = x2.shape
    dx = torch.empty_like(x2) if need_dx else None
    dgamma, dbeta = empty(C, like=x2), empty(C, like=x2)
    ag, ab = acc if acc is not None else (None, None)
    if partials is not None:
        part, nblk = partials
        pl = torch.empty(pieces(), rows, C, dtype=torch.bfloat16, device=x2.device) if planes else None
        nb_ = 4.0 * rows * C * 3 + (2.0 * pl.shape[0] * rows * C if planes else 0.0)
        ktimer.launch('bn_bwd_partials', lambda: check(lib.ha2g_bn_bwd_planes_np_partials_f32(
            dy2.data_ptr(), x2.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), _p(dx), _p(pl), pl.stride(0) if planes else 0,
            pl.shape[0] if planes else 0, dgamma.data_ptr(), dbeta.data_ptr(), rows, C, int(relu_mask), _p(ag), _p(ab), part.data_ptr(), nblk, _stream())), nb_)
        return (dx, dgamma, dbeta, pl) if planes else (dx, dgamma, dbeta)
    # algorithmic HBM bytes of the backward (bench.py roofline_bn): statistics pass reads dy and x, apply pass reads them again and writes dx
    nbytes = 4.0 * rows * C * (5 if (need_dx or planes) else 2)
    if planes:
        pl = torch.empty(pieces(), rows, C, dtype=torch.bfloat16, device=x2.device)       # piece planes [np][rows][C]
        ktimer.launch('bn_bwd', lambda: check(lib.ha2g_bn_bwd_planes_np_f32(
            dy2.data_ptr(), x2.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), _p(dx), pl.data_ptr(), pl.stride(0), pl.shape[0],
            dgamma.data_ptr(), dbeta.data_ptr(), rows, C, int(relu_mask), _p(ag), _p(ab), workspace(x2.device).data_ptr(), _stream())),
            nbytes + (4.0 * rows * C if need_dx else 0.0) + 2.0 * (pl.shape[0] - 2) * rows * C)
        return dx, dgamma, dbeta, pl
    ktimer.launch('bn_bwd', lambda: check(lib.ha2g_bn_bwd_f32(
        dy2.data_ptr(), x2.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), _p(dx), dgamma.data_ptr(), dbeta.data_ptr(), rows, C,
        int(relu_mask), _p(ag), _p(ab), workspace(x2.device).data_ptr(), _stream())), nbytes)
    return dx, dgamma, dbeta


def pieces():
    """bf16 pieces per operand of the split backward products in the current arithmetic mode (ha2g_gemm_set_mode bit 6): 3 = fp32-class
    (x = p0 + p1 + p2, all 24 mantissa bits; the default), 2 = the round-3 hi / lo split."""
    return 3 if lib.ha2g_gemm_bwd_pieces() == 3 else 2


def to_planes(x, np_=None):
    """fp32 tensor -> its bf16 piece planes, one tensor [np, *x.shape]: p0 = bf16(x), p1 = bf16(x - p0) (, p2 = bf16(x - p0 - p1)) -- the operand
    split of the split-bf16 products; np = pieces() unless given.  With np = 2, `hi, lo = to_planes(x)` unpacks the round-3 planes."""
    x = _f32c(x)
    pl = torch.empty((np_ or pieces(),) + tuple(x.shape), dtype=torch.bfloat16, device=x.device)
    check(lib.ha2g_f32_to_planes_np(x.data_ptr(), pl.data_ptr(), pl.stride(0), pl.shape[0], x.numel(), _stream()))
    return pl


class BatchNormFunction(torch.autograd.Function):
    """Train-mode BatchNorm over the last dim of a contiguous [..., C] tensor, optional fused LeakyReLU."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, act):
        x = _f32c(x.contiguous())
        x2 = x.view(-1, x.shape[-1])
        mean, invstd = bn_stats(x2, running_mean, running_var, momentum, eps)
        y = bn_apply(x2, mean, invstd, gamma, beta, act)
        _tap_act('batch_norm', y.view(x.shape), act)
        ctx.act = act
        ctx.save_for_backward(x2, mean, invstd, gamma, y if act != ACT_NONE else None)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, invstd, gamma, y = ctx.saved_tensors
        dy2 = act_bwd(dy.reshape(x2.shape), y, ctx.act)
        dx, dg, db = bn_bwd(dy2, x2, mean, invstd, gamma)
        return dx.view(dy.shape), dg, db, None, None, None, None, None


def batch_norm_train(x, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5, act=ACT_NONE):
    return BatchNormFunction.apply(x, gamma, beta, running_mean, running_var, momentum, eps, act)


def batch_norm_eval(x, gamma, beta, running_mean, running_var, eps=1e-5, act=ACT_NONE):
    """Inference BatchNorm (module.eval()): normalise with the running statistics; forward only."""
    if torch.is_grad_enabled() and (x.requires_grad or gamma.requires_grad and False):
        raise NotImplementedError('ha2g_amd: eval-mode BatchNorm is forward-only; wrap inference in torch.no_grad()')
    x = _f32c(x.contiguous())
    invstd = eltwise(OP_RSQRT_EPS, running_var, alpha=eps)
    return bn_apply(x.view(-1, x.shape[-1]), running_mean, invstd, gamma, beta, act).view(x.shape)


# ------------------------------------------------------------------------------------------------
# bidirectional multi-layer GRU
# ------------------------------------------------------------------------------------------------

def gru_supported(H):
    return bool(lib.ha2g_gru_supported_hidden(H))


USE_GRU_CLUSTER = True        # H = 300 forward recurrence on the 5-workgroup-cluster kernel (gru_cluster.hip)
_cluster_bufs = {}


def _cluster_scratch(device):
    key = (device.type, device.index)
    if key not in _cluster_bufs:
        n = lib.ha2g_gru_cluster_workspace_bytes()
        _cluster_bufs[key] = (torch.zeros(n // 8 + 8, dtype=torch.int64, device=device), torch.zeros(1, dtype=torch.int32, device=device))
    return _cluster_bufs[key]


def gru_cluster_error(device):
    """1 if any cluster hand-off timed out since start-up (synchronises; tests only)."""
    return int(_cluster_scratch(device)[1].item())


def gru_cluster_error_tensor(device):
    """The device int32 error word of the cluster GRU kernels (None while no cluster launch has happened on `device`):
    the train step appends it to its packed scalar read-back and raises when it is non-zero."""
    b = _cluster_bufs.get((device.type, device.index))
    return None if b is None else b[1]


class Ha2gClusterError(RuntimeError):
    pass


def _gru_layer_bwd(dy, y, rs, pkt, dg, hp, B, T, H, st, device, pk3t=None):
    if pk3t is not None:
        xch, err = _cluster_scratch(device)
        check(lib.ha2g_gru_layer_bwd_cluster3(dy.data_ptr(), y.data_ptr(), rs.data_ptr(), pk3t.data_ptr(), dg.data_ptr(), hp.data_ptr(), xch.data_ptr(),
                                              err.data_ptr(), B, T, H, st))
    elif USE_GRU_CLUSTER and lib.ha2g_gru_cluster_supported(H) and T <= lib.ha2g_gru_cluster_max_steps():
        xch, err = _cluster_scratch(device)
        check(lib.ha2g_gru_layer_bwd_cluster(dy.data_ptr(), y.data_ptr(), rs.data_ptr(), pkt.data_ptr(), dg.data_ptr(), hp.data_ptr(), xch.data_ptr(),
                                             err.data_ptr(), B, T, H, st))
    else:
        check(lib.ha2g_gru_layer_bwd(dy.data_ptr(), y.data_ptr(), rs.data_ptr(), pkt.data_ptr(), dg.data_ptr(), hp.data_ptr(), B, T, H, st))


GRU_FWD3 = True      # H = 300 forward recurrence on three bf16 pieces (fp32-class) in the default mode


def gru_fwd3_active(H, T):
    """the three-piece cluster forward serves this layer: cluster kernels usable and the arithmetic mode is the fp32-class default (mode 0 =
    every product on the fp32 MFMA and mode 6 keep the fp32 chain of rounds 1-3)"""
    return (GRU_FWD3 and USE_GRU_CLUSTER and bool(lib.ha2g_gru_cluster_supported(H)) and T <= lib.ha2g_gru_cluster_max_steps()
            and lib.ha2g_gemm_bwd_pieces() == 3)


def _gru_layer_fwd(gi, pk, bf, br, y, rs, B, T, H, st, device, pk3=None):
    if pk3 is not None:
        xch, err = _cluster_scratch(device)
        check(lib.ha2g_gru_layer_fwd_cluster3(gi.data_ptr(), pk3.data_ptr(), bf.data_ptr(), br.data_ptr(), y.data_ptr(), _p(rs),
                                              xch.data_ptr(), err.data_ptr(), B, T, H, st))
    elif USE_GRU_CLUSTER and lib.ha2g_gru_cluster_supported(H) and T <= lib.ha2g_gru_cluster_max_steps():
        xch, err = _cluster_scratch(device)
        check(lib.ha2g_gru_layer_fwd_cluster(gi.data_ptr(), pk.data_ptr(), bf.data_ptr(), br.data_ptr(), y.data_ptr(), _p(rs),
                                             xch.data_ptr(), err.data_ptr(), B, T, H, st))
    else:
        check(lib.ha2g_gru_layer_fwd(gi.data_ptr(), pk.data_ptr(), bf.data_ptr(), br.data_ptr(), y.data_ptr(), _p(rs), B, T, H, st))


GRU_MERGE_DIRS = True    # both directions' input projections as one plane GEMM
GRU_STACK_PREP = True    # ... whose weight operands are split for the whole stack in one launch (forward: [6H][K] planes, backward: their transposes)
PACK_MULTI = True      # one W_hh pack launch per GRU stack instead of two per layer


def _pack3_multi(whh, L, H, transposed, dev, st):
    """[L, 2, packed3 bytes]: the three-piece images (ha2g_gru_pack_whh3 / _whh3t) of the 2 L recurrent matrices from ONE launch"""
    import numpy as np
    n3 = lib.ha2g_gru_packed3_bytes()
    out = torch.empty(L, 2, n3, dtype=torch.uint8, device=dev)
    wl = [t.contiguous() for t in whh]
    wp_ = np.array([t.data_ptr() for t in wl], np.int64)
    op_ = np.array([out[l, d].data_ptr() for l in range(L) for d in range(2)], np.int64)
    check(lib.ha2g_gru_pack_whh3_multi(wp_.ctypes.data, op_.ctypes.data, 2 * L, H, int(transposed), st))
    return out


GRU_PREFETCH = False      # (measured: 34.66 / 34.77 / 34.49 with, 34.71 / 34.48 / 34.23 ms without -- nothing; off)  the weight images a GRU stack's forward AND backward need (W_hh packs, W_ih piece planes and their transposes) prepared at the start
                          # of the step on the side stream, beside the audio tower's forward (prefetch_gru; A/B switch)
_GRU_PREP = {}            # data_ptr of weight_hh_l0 -> _GruPrep of the CURRENT step (rng.step_token)


class _GruPrep:
    __slots__ = ('token', 'H', 'L', 'T', 'rows', 'fwd', 'bwd')


def _gru_prep_fwd(weights, H, L, T, rows, dev, st):
    """(pk_all, pk3_all, wih_planes, bcat_all): what BiGRUFunction.forward builds from the weights alone before its first product"""
    npk = lib.ha2g_gru_packed_floats(H)
    pk_all = None
    if PACK_MULTI and 2 * L <= 16:
        # all layers' / directions' W_hh images from ONE launch at the start of the stack (two per layer sat between the recurrences)
        import numpy as np
        pk_all = torch.empty(L, 4, npk, dtype=torch.float32, device=dev)
        wl = [weights[8 * l + 4 * d + 1].contiguous() for l in range(L) for d in range(2)]
        wp_ = np.array([t.data_ptr() for t in wl], np.int64)
        pf_ = np.array([pk_all[l, d].data_ptr() for l in range(L) for d in range(2)], np.int64)
        pb_ = np.array([pk_all[l, 2 + d].data_ptr() for l in range(L) for d in range(2)], np.int64)
        check(lib.ha2g_gru_pack_whh_multi(wp_.ctypes.data, pf_.ctypes.data, pb_.ctypes.data, 2 * L, H, st))
    pk3_all = None
    if PACK_MULTI and 2 * L <= 16 and gru_fwd3_active(H, T):
        # ... and their three-piece images (bf16 A fragments of the cluster kernel) likewise: one launch instead of two per layer
        pk3_all = _pack3_multi([weights[8 * l + 4 * d + 1] for l in range(L) for d in range(2)], L, H, False, dev, st)
    # layers 1 .. L-1 (input = 2H columns): the merged operands [W_ih; W_ih_reverse] of ALL of them as piece planes from ONE launch, their merged biases
    # from one cat (a cat of the weights, a cat of the biases and a split launch per layer before: round 6)
    wih_planes = bcat_all = None
    Kp2 = (2 * H + 31) // 32 * 32
    if (GRU_STACK_PREP and GRU_MERGE_DIRS and L > 1 and dev.type == 'cuda' and 2 * H >= 256 and 2.0 * rows * 6 * H * 2 * H >= PLANE_GEMM_MIN_FLOP
            and PLANE_GEMM and lib.ha2g_gemm_bwd_pieces() == 3 and 2 * (L - 1) <= 16
            and all(weights[8 * l + 4 * d].is_contiguous() and weights[8 * l + 4 * d].data_ptr() % 16 == 0 for l in range(1, L) for d in range(2))):
        wih_planes = torch.empty(L - 1, 3, 6 * H, Kp2, dtype=torch.bfloat16, device=dev)
        mats = [weights[8 * l + 4 * d] for l in range(1, L) for d in range(2)]
        wins = [wih_planes[l - 1, 0, 3 * H * d].data_ptr() for l in range(1, L) for d in range(2)]
        split2d_multi(mats, wins, wih_planes.stride(1), Kp2, [Kp2] * len(mats), False)
        bcat_all = torch.cat([weights[8 * l + 4 * d + 2] for l in range(1, L) for d in range(2)]).view(L - 1, 6 * H)
    return pk_all, pk3_all, wih_planes, bcat_all


def _gru_prep_bwd(weights, H, L, T, dev, st, planes):
    """(wih_t_planes, pk3t_all): what BiGRUFunction.backward builds from the weights alone; planes = the forward ran layers >= 1 on piece planes"""
    wih_t_planes = None
    if planes:
        # dX = dg[:, :6H] @ [W_ih; W_ih_reverse]: the B operand's planes hold its transpose [K][6H -> Kp6]; both directions of every layer >= 1 from one launch
        Kp6 = (6 * H + 31) // 32 * 32
        wih_t_planes = torch.empty(L - 1, 3, 2 * H, Kp6, dtype=torch.bfloat16, device=dev)
        mats = [weights[8 * l + 4 * d] for l in range(1, L) for d in range(2)]
        wins = [wih_t_planes[l - 1, 0, 0, 3 * H * d].data_ptr() for l in range(1, L) for d in range(2)]
        split2d_multi(mats, wins, wih_t_planes.stride(1), Kp6, [3 * H if d == 0 else Kp6 - 3 * H for l in range(1, L) for d in range(2)], True)
    pk3t_all = None
    if PACK_MULTI and 2 * L <= 16 and gru_fwd3_active(H, T):      # the transposed three-piece W_hh images of every layer: one launch
        pk3t_all = _pack3_multi([weights[8 * l + 4 * d + 1] for l in range(L) for d in range(2)], L, H, True, dev, st)
    return wih_t_planes, pk3t_all


def prefetch_gru(weights, H, T, rows, consumer_stream=None):
    """Prepare the weight images of one GRU stack for THIS step's forward (at `rows` = B * T rows) and backward; call inside side.section() at the start of
    the step (the weights do not change until the optimizers run) and side.join() before the stack's forward.  A forward whose shapes differ, or a later
    step, does not find the entry and prepares in line as before."""
    weights = list(weights)
    L = len(weights) // 8
    dev = weights[0].device
    st = _stream()
    p = _GruPrep()
    p.token, p.H, p.L, p.T, p.rows = rng.step_token, H, L, T, rows
    p.fwd = _gru_prep_fwd(weights, H, L, T, rows, dev, st)
    p.bwd = _gru_prep_bwd(weights, H, L, T, dev, st, p.fwd[2] is not None)
    if consumer_stream is not None:                       # allocated under the side stream, read by the consumer's: the caching allocator must know
        for t in p.fwd + p.bwd:
            if t is not None:
                t.record_stream(consumer_stream)
    _GRU_PREP[weights[1].data_ptr()] = p


def _gru_prep_lookup(weights, H, L, T, rows, dev):
    p = _GRU_PREP.get(weights[1].data_ptr()) if (GRU_PREFETCH and _GRU_PREP) else None
    if p is None or p.token != rng.step_token or (p.H, p.L, p.T) != (H, L, T) or (rows is not None and p.rows != rows):
        return None
    return p


def gru_prep_clear():
    _GRU_PREP.clear()


class BiGRUFunction(torch.autograd.Function):
    """Stacked bidirectional GRU (batch_first, h0 = 0).  forward(x, masks, H, grad_slice, *weights): weights in torch
    `_flat_weights` order (per layer, per direction: w_ih, w_hh, b_ih, b_hh); masks = tuple of pre-scaled
    dropout masks for the outputs of layers 0..L-2 (or None); grad_slice = (first_row, n_rows) promises that only
    those batch rows receive a non-zero output gradient (fused-chain step), so BPTT and the weight-gradient GEMMs
    run on that slice alone."""

    @staticmethod
    def forward(ctx, x, masks, H, grad_slice, *weights):
        B, T, _ = x.shape
        L = len(weights) // 8
        dev = x.device
        st = _stream()
        npk = lib.ha2g_gru_packed_floats(H)
        need_grad = any(w.requires_grad for w in weights) or x.requires_grad
        saved = []
        inp = x.contiguous()
        packs = []
        wcats = [None] * L
        pre = _gru_prep_lookup(weights, H, L, T, B * T, dev)
        if pre is not None:                               # prepared on the side stream while the audio tower ran (prefetch_gru, round 6)
            pk_all, pk3_all, wih_planes, bcat_all = pre.fwd
        else:
            pk_all, pk3_all, wih_planes, bcat_all = _gru_prep_fwd(weights, H, L, T, B * T, dev, st)
        for l in range(L):
            w = [t.contiguous() for t in weights[8 * l:8 * l + 8]]
            K = inp.shape[2]
            gi = torch.empty(B * T, 6 * H, dtype=torch.float32, device=dev)
            x2 = inp.view(B * T, K)
            tkey = 'gemm_gi' if (H == 300 and l > 0) else 'gemm_gi_other'
            fl = 2.0 * B * T * K * 3 * H
            if wih_planes is not None and l > 0 and K == 2 * H and x2.data_ptr() % 16 == 0:
                ktimer.launch(tkey, lambda: _gemm_planes_pre(x2, wih_planes[l - 1], B * T, 6 * H, K, gi, bias=bcat_all[l - 1]), 2 * fl)
                wcats[l] = 'planes'                        # the backward builds the transposed planes of the stack in one launch too
            elif GRU_MERGE_DIRS and _plane_gemm_ok(x2, w[0], B * T, 6 * H, K, False, True, 1.0, ACT_NONE, gi) and K >= 256:
                # both directions' input projections as ONE product [rows, K] x [6H, K]^T on the plane GEMM: twice the columns per launch fill the
                # chip's rounds (291 -> 181 us at 13056 x 1800 x 600 incl. the operand splits, profiles/r04_plane_gemm_bench.txt)
                wcat, bcat = torch.cat((w[0], w[4])), torch.cat((w[2], w[6]))
                ktimer.launch(tkey, lambda: gemm(x2, wcat, transb=True, out=gi, bias=bcat), 2 * fl)
                wcats[l] = wcat                            # [6H, K]: the backward's dX = dg[:, :6H] @ wcat is one product too
            else:
                ktimer.launch(tkey, lambda: gemm(x2, w[0], transb=True, out=gi[:, :3 * H], bias=w[2]), fl)
                ktimer.launch(tkey, lambda: gemm(x2, w[4], transb=True, out=gi[:, 3 * H:], bias=w[6]), fl)
            if pk_all is not None:
                pk = pk_all[l]
            else:
                pk = torch.empty(4, npk, dtype=torch.float32, device=dev)     # [fwd-form f, r | bwd-form f, r]
                check(lib.ha2g_gru_pack_whh(w[1].data_ptr(), pk[0].data_ptr(), pk[2].data_ptr(), H, st))
                check(lib.ha2g_gru_pack_whh(w[5].data_ptr(), pk[1].data_ptr(), pk[3].data_ptr(), H, st))
            y = torch.empty(B, T, 2 * H, dtype=torch.float32, device=dev)
            rs = torch.empty(B, T, 2, 4, H, dtype=torch.float32, device=dev) if need_grad else None
            pk3 = None
            if pk3_all is not None:
                pk3 = pk3_all[l]
            elif gru_fwd3_active(H, T):                    # three-piece W_hh images of both directions (bf16 A fragments)
                n3 = lib.ha2g_gru_packed3_bytes()
                pk3 = torch.empty(2, n3, dtype=torch.uint8, device=dev)
                check(lib.ha2g_gru_pack_whh3(w[1].data_ptr(), pk3[0].data_ptr(), H, st))
                check(lib.ha2g_gru_pack_whh3(w[5].data_ptr(), pk3[1].data_ptr(), H, st))
            ktimer.launch('gru_layer_fwd' if (H == 300 and l > 0) else 'gru_layer_fwd_other',
                          lambda: _gru_layer_fwd(gi, pk, w[3], w[7], y, rs, B, T, H, st, dev, pk3), B)
            saved.append((inp, y, rs))
            packs.append(pk)
            inp = y
            if masks is not None and l < L - 1 and masks[l] is not None:
                inp = _drop_apply(y, masks[l]) if isinstance(masks[l], DropSpec) else eltwise(OP_MUL, y, masks[l])
        ctx.H, ctx.L, ctx.masks, ctx.grad_slice = H, L, masks, grad_slice
        ctx.saved_bufs = saved
        ctx.packs = packs
        ctx.wcats = wcats
        ctx.save_for_backward(*weights)
        return inp

    @staticmethod
    def backward(ctx, dy):
        H, L, masks = ctx.H, ctx.L, ctx.masks
        weights = ctx.saved_tensors
        st = _stream()
        dy = dy.contiguous()
        Bfull = dy.shape[0]
        sl = slice(None)
        if ctx.grad_slice is not None and ctx.grad_slice[1] < Bfull:
            sl = slice(ctx.grad_slice[0], ctx.grad_slice[0] + ctx.grad_slice[1])
            dy = dy[sl]                                   # leading-dim slice: still contiguous
        B, T, _ = dy.shape
        dev = dy.device
        grads = [None] * (8 * L)
        keep = []
        fused_b = []
        all_tgs = []                                       # the .grad buffers the side-stream sections below accumulate into (SideStream.touch)
        need_planes = any(isinstance(c, str) for c in ctx.wcats)
        pre = _gru_prep_lookup(weights, H, L, T, None, dev)
        if pre is not None and (pre.bwd[0] is not None) == need_planes:
            wih_t_planes, pk3t_all = pre.bwd                  # prepared at the start of the step (prefetch_gru)
        else:
            wih_t_planes, pk3t_all = _gru_prep_bwd(weights, H, L, T, dev, st, need_planes)
        for l in range(L - 1, -1, -1):
            inp, y, rs = (t[sl] for t in ctx.saved_bufs[l])
            w = weights[8 * l:8 * l + 8]
            if masks is not None and l < L - 1 and masks[l] is not None:
                if isinstance(masks[l], DropSpec):         # the mask of rows [sl] re-drawn: element offset of the slice inside the [Bfull, T, 2H] tensor
                    spec = masks[l]
                    if spec.token != rng.step_token:
                        raise RuntimeError('ha2g_amd dropout: backward after the RNG step advanced (rng.end_step()): the mask cannot be re-drawn')
                    dyc = dy.contiguous()
                    out = torch.empty_like(dyc)
                    off = (sl.start or 0) * T * 2 * H
                    check(lib.ha2g_dropout_slice_f32(dyc.data_ptr(), out.data_ptr(), dyc.numel(), off, spec.p, rng.state.data_ptr(), spec.call, st))
                    dy = out
                else:
                    dy = eltwise(OP_MUL, dy, masks[l][sl])
            dg = torch.empty(B * T, 8 * H, dtype=torch.float32, device=dev)        # [(r z n) fwd | (r z n) rev | hn fwd | hn rev]
            # h_prev per direction (forward dir: y[t-1], reverse dir: y[t+1], zero at the sequence ends) is written by the BPTT kernel
            hp = torch.empty(B, T, 2 * H, dtype=torch.float32, device=dev)
            pk3t = None
            if pk3t_all is not None:
                pk3t = pk3t_all[l]
            elif gru_fwd3_active(H, T):                    # the BPTT chain on three pieces too: transposed W_hh images of both directions
                n3 = lib.ha2g_gru_packed3_bytes()
                pk3t = torch.empty(2, n3, dtype=torch.uint8, device=dev)
                check(lib.ha2g_gru_pack_whh3t(w[1].data_ptr(), pk3t[0].data_ptr(), H, st))
                check(lib.ha2g_gru_pack_whh3t(w[5].data_ptr(), pk3t[1].data_ptr(), H, st))
            ktimer.launch('gru_layer_bwd' if H == 300 else 'gru_layer_bwd_other',
                          lambda: _gru_layer_bwd(dy, y, rs, ctx.packs[l][2], dg, hp, B, T, H, st, dev, pk3t), B)
            K = inp.shape[2]
            x2 = inp.view(B * T, K)
            hp2 = hp.view(B * T, 2 * H)
            need_dx = l > 0 or ctx.needs_input_grad[0]
            dx = torch.empty(B * T, K, dtype=torch.float32, device=dev) if need_dx else None
            keep.append((dg, hp, x2, y, dy))                                        # side-stream readers: alive until join

            def data_gradient():
                if need_dx and isinstance(ctx.wcats[l], str):
                    _gemm_planes_pre(dg[:, :6 * H], wih_t_planes[l - 1], B * T, K, 6 * H, dx)
                elif need_dx and ctx.wcats[l] is not None:                          # critical path: dX = [dgi fwd | dgi rev] [W_ih; W_ih_reverse], ONE product
                    gemm(dg[:, :6 * H], ctx.wcats[l], out=dx)                       # (two products of K = 3H with an accumulate pass before: 2 x 121 us in the step)
                elif need_dx:
                    for d in range(2):                                              # dX (+)= dgi W_ih
                        gemm(dg[:, 3 * H * d:3 * H * d + 3 * H], w[4 * d], out=dx, beta=float(d))
            if GRU_DX_FIRST:
                # the layer's data gradient is enqueued BEFORE the side queue forks: the weight-gradient products then start when dX is done and run beside
                # the next layer's recurrence (80 workgroups, latency-bound) instead of beside the product the recurrence is waiting for
                data_gradient()
            with side.section(dev):                                                 # weight / bias gradients: off the critical path
                tgs = []
                for d in range(2):
                    tg = [_grad_target(w[4 * d + i]) for i in range(4)]
                    tgs.append([t if (t is not None and t.is_contiguous()) else None for t in tg])
                all_tgs += tgs
                if FUSE_BIAS_GRAD and GROUP_GRU_WGRAD and all(t is not None for tg in tgs for t in tg):
                    # every target is an installed .grad buffer: the two directions' weight-gradient GEMMs have one shape each -> three grouped
                    # launches (dW_ih, dW_hh rows r z, dW_hh rows n) instead of six, bias gradients riding on them
                    # (raw addresses of the column / row blocks: no slice views -- this backward is on the host-bound stretch of the step)
                    assert x2.stride() == (K, 1) and all(tgs[d][0].shape == (3 * H, K) and tgs[d][1].shape == (3 * H, H) for d in range(2))
                    pdg, px, ph = dg.data_ptr(), x2.data_ptr(), hp2.data_ptr()
                    pw = [[tgs[d][i].data_ptr() for i in range(4)] for d in range(2)]
                    gemm_grouped_raw(dev, 2, True, False, 3 * H, K, B * T, [pdg + 12 * H * d for d in range(2)], 8 * H, [px, px], K,
                                     [pw[d][0] for d in range(2)], K, 1.0, [pw[d][2] for d in range(2)], 1.0)
                    gemm_grouped_raw(dev, 2, True, False, 2 * H, H, B * T, [pdg + 12 * H * d for d in range(2)], 8 * H, [ph + 4 * H * d for d in range(2)], 2 * H,
                                     [pw[d][1] for d in range(2)], H, 1.0, [pw[d][3] for d in range(2)], 1.0)
                    gemm_grouped_raw(dev, 2, True, False, H, H, B * T, [pdg + 4 * (6 * H + H * d) for d in range(2)], 8 * H, [ph + 4 * H * d for d in range(2)], 2 * H,
                                     [pw[d][1] + 8 * H * H for d in range(2)], H, 1.0, [pw[d][3] + 8 * H for d in range(2)], 1.0)
                    fused_b += [True, True]
                else:
                    for d in range(2):
                        o = 3 * H * d
                        dgi = dg[:, o:o + 3 * H]
                        tg = tgs[d]
                        # bias gradients ride on the weight-gradient launches when every target is an installed .grad buffer:
                        # b_ih <- columns [r z n] of dgi, b_hh <- columns [r z] and [hn] (the same dY tiles the three GEMMs stage)
                        fb = FUSE_BIAS_GRAD and all(t is not None for t in tg)
                        cs3 = [dict(colsum_out=c, colsum_beta=1.0) for c in (tg[2], tg[3][:2 * H], tg[3][2 * H:])] if fb else [{}, {}, {}]
                        fused_b.append(fb)
                        if tg[0] is not None:
                            gemm(dgi, x2, transa=True, out=tg[0], beta=1.0, **cs3[0])   # dW_ih += dgi^T X, straight into .grad
                        else:
                            grads[8 * l + 4 * d + 0] = gemm(dgi, x2, transa=True)
                        bt = 1.0 if tg[1] is not None else 0.0
                        dwhh = tg[1] if tg[1] is not None else torch.empty(3 * H, H, dtype=torch.float32, device=dev)
                        hpd = hp2[:, d * H:(d + 1) * H]
                        gemm(dg[:, o:o + 2 * H], hpd, transa=True, out=dwhh[:2 * H], beta=bt, **cs3[1])    # rows r,z
                        gemm(dg[:, 6 * H + H * d:6 * H + H * (d + 1)], hpd, transa=True, out=dwhh[2 * H:], beta=bt, **cs3[2])  # rows n (d gh_n)
                        if tg[1] is None:
                            grads[8 * l + 4 * d + 1] = dwhh
                if not all(fused_b[-2:]):
                    assert not any(fused_b[-2:]), 'both directions of a layer share one gradient-buffer state'
                    # the four bias gradients of the layer from ONE column sum over all 8H gate-gradient columns
                    cs = colsum(dg)
                    direct = all(t[2] is not None and t[3] is not None for t in tgs)
                    outs = [(tgs[d][2], tgs[d][3]) if direct else (torch.empty(3 * H, dtype=torch.float32, device=dev),
                                                                    torch.empty(3 * H, dtype=torch.float32, device=dev)) for d in range(2)]
                    check(lib.ha2g_gru_bias_grads_f32(cs.data_ptr(), outs[0][0].data_ptr(), outs[0][1].data_ptr(), outs[1][0].data_ptr(),
                                                      outs[1][1].data_ptr(), H, 1.0 if direct else 0.0, _stream()))
                    keep.append(cs)
                    if not direct:
                        for d in range(2):
                            grads[8 * l + 4 * d + 2], grads[8 * l + 4 * d + 3] = outs[d]
            if not GRU_DX_FIRST:
                data_gradient()
            dy = dx.view(B, T, K) if need_dx else None
        if DEFER_JOIN and side.allow_defer and fused_b and all(fused_b) and all(g is None for g in grads):
            # every weight / bias gradient of the stack accumulated in place into installed .grad buffers: the main stream need not wait for the side
            # queue here (5 joins of ~75 us per step) -- the step flushes before the gradient exchange / the optimizer
            side.defer(dev, keep, [t for tg in all_tgs for t in tg])
        else:
            side.join(dev)
        if dy is not None and B != Bfull:
            full = torch.zeros(Bfull, T, dy.shape[2], dtype=torch.float32, device=dev)
            full[sl] = dy
            dy = full
        ctx.saved_bufs = ctx.wcats = None
        return (dy, None, None, None) + tuple(grads)


def bigru(x, weights, H, masks=None, grad_slice=None):
    return BiGRUFunction.apply(x, masks, H, grad_slice, *weights)


# ------------------------------------------------------------------------------------------------
# loss terms: value + unit gradient computed in forward; backward = scale by the upstream scalar
# ------------------------------------------------------------------------------------------------

def _scale_by(g_unit, gout, sign=1.0):
    return eltwise(OP_MUL_SCALAR, g_unit, gout.contiguous(), alpha=sign)


class HuberFunction(torch.autograd.Function):
    """mean Huber_beta(x - y) == smooth_l1_loss(x/beta, y/beta) * beta  (train_hierarchy.py:173-176)."""

    @staticmethod
    def forward(ctx, x, y, beta):
        x, y = _f32c(x.contiguous()), _f32c(y.contiguous())
        loss, dx = empty((), like=x), torch.empty_like(x)
        check(lib.ha2g_huber_f32(x.data_ptr(), y.data_ptr(), x.numel(), beta, loss.data_ptr(), dx.data_ptr(),
                                 workspace(x.device).data_ptr(), _stream()))
        ctx.save_for_backward(dx)
        return loss

    @staticmethod
    def backward(ctx, g):
        return _scale_by(ctx.saved_tensors[0], g), None, None


def huber(x, y, beta):
    return HuberFunction.apply(x, y, beta)


class KLDFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, logvar):
        mu, logvar = _f32c(mu.contiguous()), _f32c(logvar.contiguous())
        loss, dmu, dlv = empty((), like=mu), torch.empty_like(mu), torch.empty_like(mu)
        check(lib.ha2g_kld_f32(mu.data_ptr(), logvar.data_ptr(), mu.numel(), loss.data_ptr(), dmu.data_ptr(), dlv.data_ptr(), _stream()))
        ctx.save_for_backward(dmu, dlv)
        return loss

    @staticmethod
    def backward(ctx, g):
        dmu, dlv = ctx.saved_tensors
        return _scale_by(dmu, g), _scale_by(dlv, g)


def kld(mu, logvar):
    return KLDFunction.apply(mu, logvar)


class DivRegFunction(torch.autograd.Function):
    """train_hierarchy.py:213-222; gradient flows into `out` only (everything else is detached there)."""

    @staticmethod
    def forward(ctx, out, rnd, z, zr, beta):
        out, rnd, z, zr = (_f32c(t.contiguous()) for t in (out, rnd, z, zr))
        B = out.shape[0]
        loss, dout = empty((), like=out), torch.empty_like(out)
        check(lib.ha2g_divreg_f32(out.data_ptr(), rnd.data_ptr(), z.data_ptr(), zr.data_ptr(), B, out[0].numel(), z.shape[1], beta,
                                  loss.data_ptr(), dout.data_ptr(), workspace(out.device).data_ptr(), _stream()))
        ctx.save_for_backward(dout)
        return loss

    @staticmethod
    def backward(ctx, g):
        return _scale_by(ctx.saved_tensors[0], g), None, None, None, None


def div_reg(out, rnd, z, zr, beta=0.05):
    return DivRegFunction.apply(out, rnd, z, zr, beta)


class PhysAngleFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, mean_dir, pairs, avg, var, palm):
        import ctypes
        out = _f32c(out.contiguous())
        P = out.shape[-1]
        rows = out.numel() // P
        loss, dout = empty((), like=out), torch.empty_like(out)
        flat = [int(v) for pr in palm for v in pr]
        parr = (ctypes.c_int * max(len(flat), 1))(*flat)
        check(lib.ha2g_phys_angle_f32(out.data_ptr(), mean_dir.data_ptr(), rows, P // 3, pairs.data_ptr(), pairs.shape[0],
                                      avg.data_ptr(), var.data_ptr(), ctypes.cast(parr, ctypes.c_void_p), len(palm),
                                      loss.data_ptr(), dout.data_ptr(), workspace(out.device).data_ptr(), _stream()))
        ctx.save_for_backward(dout)
        return loss

    @staticmethod
    def backward(ctx, g):
        return _scale_by(ctx.saved_tensors[0], g), None, None, None, None, None


def phys_angle(out, mean_dir, pairs, avg, var, palm=()):
    return PhysAngleFunction.apply(out, mean_dir, pairs, avg, var, tuple(palm))


class GanLossFunction(torch.autograd.Function):
    """mode 0: -mean(log(a + 1e-8)); mode 1: -mean(log(a + 1e-8) + log(1 - b + 1e-8))  (train_hierarchy.py:128,180)."""

    @staticmethod
    def forward(ctx, mode, a, b):
        a = _f32c(a.contiguous())
        b = _f32c(b.contiguous()) if b is not None else None
        loss, da = empty((), like=a), torch.empty_like(a)
        db = torch.empty_like(a) if b is not None else None
        check(lib.ha2g_gan_loss_f32(mode, a.data_ptr(), _p(b), a.numel(), loss.data_ptr(), da.data_ptr(), _p(db), _stream()))
        ctx.save_for_backward(da, db)
        return loss

    @staticmethod
    def backward(ctx, g):
        da, db = ctx.saved_tensors
        return None, _scale_by(da, g), (_scale_by(db, g) if db is not None else None)


def gen_loss(d_out):
    return GanLossFunction.apply(0, d_out, None)


def dis_loss(real, fake):
    return GanLossFunction.apply(1, real, fake)


class ContrastiveFunction(torch.autograd.Function):
    """SoftmaxContrastiveLoss (train_hierarchy.py:54-68 / train_hierarchy_expressive.py:107-121); a, b [N,32]."""

    @staticmethod
    def forward(ctx, a, b, expressive):
        a, b = _f32c(a.contiguous()), _f32c(b.contiguous())
        N = a.shape[0]
        assert a.shape == b.shape and a.shape[1] == 32
        ws = workspace(a.device)
        assert lib.ha2g_contrastive_workspace_floats(N) <= ws.numel()
        loss, da, db = empty((), like=a), torch.empty_like(a), torch.empty_like(b)
        check(lib.ha2g_contrastive_f32(a.data_ptr(), b.data_ptr(), N, int(expressive), loss.data_ptr(), da.data_ptr(), db.data_ptr(),
                                       ws.data_ptr(), _stream()))
        ctx.save_for_backward(da, db)
        return loss

    @staticmethod
    def backward(ctx, g):
        da, db = ctx.saved_tensors
        return _scale_by(da, g), _scale_by(db, g), None


def contrastive(a, b, expressive=False):
    return ContrastiveFunction.apply(a, b, expressive)


# ------------------------------------------------------------------------------------------------
# generator input pack / hierarchy scatter / loss assembly (csrc/pack.hip)
# ------------------------------------------------------------------------------------------------

def scatter_tables(P, Pprev, scatter, device):
    """(map[P+1], inv[Pprev][2]) int32 device tables of one hierarchy level from the reference's slice assignments
    `pre_seq_k[:, n:, dst] = out_{k-1}[:, n:, src]` applied in order on the (P+1)-wide pre_seq (python slice semantics, negative
    bounds included: that is what reproduces the expressive step's one-column shift; a later assignment overwrites an earlier one)."""
    import numpy as np
    m = -np.ones(P + 1, np.int32)
    for dst, src in scatter:
        m[dst] = np.arange(Pprev, dtype=np.int32)[src]
    inv = -np.ones((max(Pprev, 1), 2), np.int32)
    for c, j in enumerate(m):
        if j >= 0:
            slot = 0 if inv[j, 0] < 0 else 1
            assert inv[j, slot] < 0, 'an output column feeds more than two pre_seq columns'
            inv[j, slot] = c
    return torch.from_numpy(m).to(device), torch.from_numpy(inv).to(device)


class PreSeqFunction(torch.autograd.Function):
    """pre_seq of one level (train_hierarchy.py:153-169): frames < n_pre carry (target, 1), later frames the coarser level's
    output through the scatter map; differentiable w.r.t. that output."""

    @staticmethod
    def forward(ctx, target_k, prev, tables, n_pre):
        target_k = _f32c(target_k.contiguous())
        R, T, P = target_k.shape
        out = torch.empty(R, T, P + 1, dtype=torch.float32, device=target_k.device)
        Pprev = 0
        if prev is not None:
            prev = _f32c(prev.contiguous())
            Pprev = prev.shape[2]
        check(lib.ha2g_pre_seq_fwd_f32(target_k.data_ptr(), _p(prev), _p(tables[0]) if prev is not None else 0, out.data_ptr(), R, T, P, Pprev,
                                       n_pre, _stream()))
        ctx.geom, ctx.tables = (R, T, P, Pprev, n_pre), tables
        return out

    @staticmethod
    def backward(ctx, d):
        R, T, P, Pprev, n_pre = ctx.geom
        if Pprev == 0 or not ctx.needs_input_grad[1]:
            return None, None, None, None
        d = _f32c(d.contiguous())
        dprev = torch.empty(R, T, Pprev, dtype=torch.float32, device=d.device)
        check(lib.ha2g_pre_seq_bwd_f32(d.data_ptr(), ctx.tables[1].data_ptr(), dprev.data_ptr(), R, T, P, Pprev, n_pre, _stream()))
        return None, dprev, None, None


def pre_seq(target_k, prev, tables, n_pre):
    return PreSeqFunction.apply(target_k, prev, tables, n_pre)


class GenConcatFunction(torch.autograd.Function):
    """in_data = cat(pre_seq, audio_feat, text_feat, z expanded over time) (model/hierarchy_net.py:121-141, input_context 'both')."""

    @staticmethod
    def forward(ctx, a, b, c, z):
        a, b, c, z = (_f32c(t.contiguous()) for t in (a, b, c, z))
        R, T, Wa = a.shape
        Wb, Wc, Wz = b.shape[2], c.shape[2], z.shape[1]
        assert b.shape[:2] == (R, T) and c.shape[:2] == (R, T) and z.shape[0] == R
        out = torch.empty(R, T, Wa + Wb + Wc + Wz, dtype=torch.float32, device=a.device)
        check(lib.ha2g_gen_concat_fwd_f32(a.data_ptr(), b.data_ptr(), c.data_ptr(), z.data_ptr(), out.data_ptr(), R, T, Wa, Wb, Wc, Wz, _stream()))
        ctx.geom = (R, T, Wa, Wb, Wc, Wz)
        return out

    @staticmethod
    def backward(ctx, d):
        R, T, Wa, Wb, Wc, Wz = ctx.geom
        d = _f32c(d.contiguous())
        need = ctx.needs_input_grad
        mk = lambda ok, *shape: torch.empty(*shape, dtype=torch.float32, device=d.device) if ok else None
        da, db, dc, dz = mk(need[0], R, T, Wa), mk(need[1], R, T, Wb), mk(need[2], R, T, Wc), mk(need[3], R, Wz)
        check(lib.ha2g_gen_concat_bwd_f32(d.data_ptr(), _p(da), _p(db), _p(dc), _p(dz), R, T, Wa, Wb, Wc, Wz, _stream()))
        return da, db, dc, dz


def gen_concat(a, b, c, z):
    return GenConcatFunction.apply(a, b, c, z)


class WeightedSumFunction(torch.autograd.Function):
    """sum_i w_i * term_i over 0-dim loss tensors in one launch (train_hierarchy.py:226-262); backward = one launch too."""

    @staticmethod
    def forward(ctx, weights, *terms):
        import ctypes
        n = len(terms)
        terms = [_f32c(t) for t in terms]
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in terms])
        ws = (ctypes.c_float * n)(*[float(w) for w in weights])
        out = torch.empty((), dtype=torch.float32, device=terms[0].device)
        check(lib.ha2g_weighted_sum_f32(ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(ws, ctypes.c_void_p), n, out.data_ptr(), _stream()))
        ctx.weights = [float(w) for w in weights]
        return out

    @staticmethod
    def backward(ctx, g):
        import ctypes
        n = len(ctx.weights)
        ws = (ctypes.c_float * n)(*ctx.weights)
        out = torch.empty(n, dtype=torch.float32, device=g.device)
        check(lib.ha2g_weighted_sum_bwd_f32(ctypes.cast(ws, ctypes.c_void_p), n, g.contiguous().data_ptr(), out.data_ptr(), _stream()))
        return (None,) + tuple(out[i] for i in range(n))


def weighted_sum(terms, weights):
    return WeightedSumFunction.apply(tuple(weights), *terms)


# ------------------------------------------------------------------------------------------------
# sparse embedding tables (csrc/sparse.hip): compact row gradients + lazy row-wise Adam
# ------------------------------------------------------------------------------------------------

class SparseTable:
    """Optimizer-side state of one embedding table [n_rows, C] that is updated row-wise (ha2g_amd.optim.FusedAdam(sparse=...)).
    `pending` collects this step's compact gradients (uniq ids, device count, summed rows) from EmbeddingFunction.backward."""

    def __init__(self, weight, opt):
        V, C = weight.shape
        dev = weight.device
        self.weight, self.opt = weight, opt
        self.m = torch.zeros(V, C, dtype=torch.float32, device=dev)
        self.v = torch.zeros(V, C, dtype=torch.float32, device=dev)
        self.last = torch.zeros(V, dtype=torch.int32, device=dev)                    # optimizer step each row is up to date with
        self.map = torch.full((V,), 2 ** 31 - 1, dtype=torch.int32, device=dev)      # scratch of ha2g_unique_tokens
        self.pending = []
        self.count_hint = None          # (device count tensor, prefetched max-over-ranks handle) of the forward whose backward is pending
        self.n_prefetch = 0             # count collectives issued since the last exchange -- equal on every rank (every rank runs every forward)

    def _run(self, ids, count, max_rows, vals):
        g = self.opt.param_groups[0]
        w = self.weight.data
        guard = gru_cluster_error_tensor(w.device)          # flagged step: no real row update (see FusedAdam.step)
        check(lib.ha2g_sparse_adam2_f32(w.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.last.data_ptr(), ids.data_ptr(), count.data_ptr(),
                                        max_rows, _p(vals), self.opt.table.data_ptr(), self.opt.step_t.data_ptr(), w.shape[1],
                                        float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), self.opt.TABLE_STEPS, float(g['lr']),
                                        _p(guard), _stream()))

    def catch_up(self, ids, count, max_rows):
        self._run(ids, count, max_rows, None)

    def prefetch_count(self, count):
        """Data parallel only.  The row exchange after the backward (ddp.gather_sparse_rows) sizes its all-gathers by the MAX over ranks of the
        number of distinct rows -- a host-side number.  The count depends on the token batch alone, so it is known at FORWARD time: the count
        all-gather and its device -> host copy are enqueued here, a whole forward + backward ahead of their use, and the exchange then reads a
        pinned word whose event fired long ago instead of stalling the launch queue on `.item()` (one stall per table and step before)."""
        from . import ddp
        self.count_hint = None
        if ddp.active():
            self.count_hint = (count, ddp.prefetch_max_count(count))
            self.n_prefetch += 1

    def merged(self):
        """This step's compact gradient as ONE (ids, count, rows) list with distinct ids (several backward passes / ranks are merged
        by running the id lists through the same compaction + deterministic row summation the embedding backward uses)."""
        if len(self.pending) == 1:
            return self.pending[0]
        ids = torch.cat([torch.where(torch.arange(u.numel(), device=u.device) < c, u, torch.zeros_like(u)) for u, c, _ in self.pending])
        rows = torch.cat([v * (torch.arange(v.shape[0], device=v.device) < c).unsqueeze(1) for _, c, v in self.pending])
        return merge_rows(ids, rows, self.map)

    def step(self):
        if not self.pending:
            return
        ids, count, vals = self.merged()
        self._run(ids, count, ids.numel(), vals)
        self.pending = []

    def sync(self):
        """Bring EVERY row up to date (before reading the table outside the embedding forward: checkpoints, evaluation dumps)."""
        V = self.weight.shape[0]
        ids = torch.empty(V, dtype=torch.int64, device=self.weight.device)
        count = torch.empty(1, dtype=torch.int32, device=self.weight.device)
        check(lib.ha2g_iota_ids(ids.data_ptr(), count.data_ptr(), V, _stream()))
        self.catch_up(ids, count, V)


def merge_rows(ids, rows, map_scratch):
    """(ids [n] with repeats -- entries that carry nothing must be id 0 with a zero row --, rows [n, C]) -> (uniq, count, summed rows)."""
    n, C = rows.shape
    dev = rows.device
    ids = ids.contiguous()
    uniq = torch.empty(n + 1, dtype=torch.int64, device=dev)
    remap = torch.empty(n, dtype=torch.int64, device=dev)
    cpos = torch.empty(n, dtype=torch.int32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    check(lib.ha2g_unique_tokens(ids.data_ptr(), n, map_scratch.data_ptr(), cpos.data_ptr(), uniq.data_ptr(), remap.data_ptr(), count.data_ptr(), _stream()))
    vals = torch.zeros(n + 1, C, dtype=torch.float32, device=dev)
    check(lib.ha2g_embedding_bwd_f32(remap.data_ptr(), rows.contiguous().data_ptr(), vals.data_ptr(), n, C, 0, workspace(dev).data_ptr(), _stream()))
    return uniq, count, vals

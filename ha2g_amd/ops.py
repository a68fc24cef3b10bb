"""Python face of the HIP kernels: thin functional wrappers (raw pointers + current stream through the C-ABI)
and the torch.autograd.Function classes that make `loss.backward()` at the reference's call sites run the
hand-written backward kernels.  torch is used for device memory, streams and autograd bookkeeping only.
"""
import torch

from ._lib import check, lib

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_SIGMOID = 0, 1, 2, 3

_WS_BYTES = 96 << 20
_ws = {}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def workspace(device):
    """One persistent split-K / scratch workspace per device (all ops of a step run on one stream)."""
    key = (device.type, device.index)
    if key not in _ws:
        _ws[key] = torch.empty(_WS_BYTES // 4, dtype=torch.float32, device=device)
    return _ws[key]


def _p(t):
    return 0 if t is None else t.data_ptr()


def _chk2d(t):
    assert t.dim() == 2 and t.stride(1) == 1 and t.dtype == torch.float32 and t.is_cuda, (t.shape, t.stride(), t.dtype)


def gemm(a, b, transa=False, transb=False, out=None, alpha=1.0, beta=0.0, bias=None, act=ACT_NONE):
    """out[M,N] = act(alpha * op(a) @ op(b) + beta*out + bias).  a, b, out: 2-D fp32 CUDA tensors with unit
    inner stride (row stride free, so column slices of wider buffers work)."""
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
    check(lib.ha2g_gemm_f32(int(transa), int(transb), M, N, K, alpha, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0),
                            beta, out.data_ptr(), out.stride(0), _p(bias), act, ws.data_ptr(), ws.numel() * 4, _stream()))
    return out


def colsum(x, out=None, beta=0.0):
    _chk2d(x)
    if out is None:
        out = torch.empty(x.shape[1], dtype=torch.float32, device=x.device)
    check(lib.ha2g_colsum_f32(x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], out.data_ptr(), beta, _stream()))
    return out


# ------------------------------------------------------------------------------------------------
# bidirectional multi-layer GRU
# ------------------------------------------------------------------------------------------------

def gru_supported(H):
    return bool(lib.ha2g_gru_supported_hidden(H))


class BiGRUFunction(torch.autograd.Function):
    """Stacked bidirectional GRU (batch_first, h0 = 0).  forward(x, masks, H, *weights): weights in torch
    `_flat_weights` order (per layer, per direction: w_ih, w_hh, b_ih, b_hh); masks = tuple of pre-scaled
    dropout masks for the outputs of layers 0..L-2 (or None)."""

    @staticmethod
    def forward(ctx, x, masks, H, *weights):
        B, T, _ = x.shape
        L = len(weights) // 8
        dev = x.device
        st = _stream()
        npk = lib.ha2g_gru_packed_floats(H)
        need_grad = any(w.requires_grad for w in weights) or x.requires_grad
        saved = []
        inp = x.contiguous()
        packs = []
        for l in range(L):
            w = weights[8 * l:8 * l + 8]
            K = inp.shape[2]
            gi = torch.empty(B * T, 6 * H, dtype=torch.float32, device=dev)
            x2 = inp.view(B * T, K)
            gemm(x2, w[0], transb=True, out=gi[:, :3 * H], bias=w[2])
            gemm(x2, w[4], transb=True, out=gi[:, 3 * H:], bias=w[6])
            pk = torch.empty(4, npk, dtype=torch.float32, device=dev)     # [fwd-form f, r | bwd-form f, r]
            check(lib.ha2g_gru_pack_whh(w[1].data_ptr(), pk[0].data_ptr(), pk[2].data_ptr(), H, st))
            check(lib.ha2g_gru_pack_whh(w[5].data_ptr(), pk[1].data_ptr(), pk[3].data_ptr(), H, st))
            y = torch.empty(B, T, 2 * H, dtype=torch.float32, device=dev)
            rs = torch.empty(B, T, 2, 4, H, dtype=torch.float32, device=dev) if need_grad else None
            check(lib.ha2g_gru_layer_fwd(gi.data_ptr(), pk.data_ptr(), w[3].data_ptr(), w[7].data_ptr(), y.data_ptr(), _p(rs),
                                         B, T, H, st))
            saved.append((inp, y, rs))
            packs.append(pk)
            inp = y
            if masks is not None and l < L - 1 and masks[l] is not None:
                inp = y * masks[l]
        ctx.H, ctx.L, ctx.masks = H, L, masks
        ctx.saved_bufs = saved
        ctx.packs = packs
        ctx.save_for_backward(*weights)
        return inp

    @staticmethod
    def backward(ctx, dy):
        H, L, masks = ctx.H, ctx.L, ctx.masks
        weights = ctx.saved_tensors
        st = _stream()
        dy = dy.contiguous()
        B, T, _ = dy.shape
        dev = dy.device
        grads = [None] * (8 * L)
        for l in range(L - 1, -1, -1):
            inp, y, rs = ctx.saved_bufs[l]
            w = weights[8 * l:8 * l + 8]
            if masks is not None and l < L - 1 and masks[l] is not None:
                dy = dy * masks[l]
            dg = torch.empty(B * T, 8 * H, dtype=torch.float32, device=dev)        # [dir][r z n hn]
            check(lib.ha2g_gru_layer_bwd(dy.data_ptr(), y.data_ptr(), rs.data_ptr(), ctx.packs[l][2].data_ptr(), dg.data_ptr(),
                                         B, T, H, st))
            K = inp.shape[2]
            x2 = inp.view(B * T, K)
            # h_prev per direction: forward dir sees y[t-1], reverse dir sees y[t+1]; zero at the sequence ends
            hp = torch.zeros(B, T, 2 * H, dtype=torch.float32, device=dev)
            hp[:, 1:, :H] = y[:, :-1, :H]
            hp[:, :-1, H:] = y[:, 1:, H:]
            hp2 = hp.view(B * T, 2 * H)
            dx = torch.empty(B * T, K, dtype=torch.float32, device=dev)
            for d in range(2):
                o = 4 * H * d
                dgi = dg[:, o:o + 3 * H]
                w_ih = w[4 * d]
                gemm(dgi, w_ih, out=dx, beta=float(d))                              # dX (+)= dgi W_ih
                grads[8 * l + 4 * d + 0] = gemm(dgi, x2, transa=True)               # dW_ih = dgi^T X
                dwhh = torch.empty(3 * H, H, dtype=torch.float32, device=dev)
                hpd = hp2[:, d * H:(d + 1) * H]
                gemm(dg[:, o:o + 2 * H], hpd, transa=True, out=dwhh[:2 * H])        # rows r,z
                gemm(dg[:, o + 3 * H:o + 4 * H], hpd, transa=True, out=dwhh[2 * H:])  # rows n (d gh_n)
                grads[8 * l + 4 * d + 1] = dwhh
                grads[8 * l + 4 * d + 2] = colsum(dgi)
                dbhh = torch.empty(3 * H, dtype=torch.float32, device=dev)
                colsum(dg[:, o:o + 2 * H], out=dbhh[:2 * H])
                colsum(dg[:, o + 3 * H:o + 4 * H], out=dbhh[2 * H:])
                grads[8 * l + 4 * d + 3] = dbhh
            dy = dx.view(B, T, K)
        ctx.saved_bufs = None
        return (dy, None, None) + tuple(grads)


def bigru(x, weights, H, masks=None):
    return BiGRUFunction.apply(x, masks, H, *weights)

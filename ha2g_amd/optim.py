"""Fused Adam over flat parameter/gradient buffers (HIP kernel ha2g_adam_f32) with torch.optim.Adam semantics as the
reference sets it up (scripts/train.py:155-170: one optimizer per module, lr 5e-4 (D: x0.2), betas (0.5, 0.999)).

Construction re-points every parameter's storage into one flat fp32 buffer (layout-preserving, so channels_last conv
weights stay OHWI) and installs `.grad` views into a matching flat gradient buffer: the optimizer step is then ONE
HBM-streaming kernel, and data-parallel gradient exchange is ONE RCCL all-reduce per module (ha2g_amd/ddp.py).
"""
import torch

from ._lib import check, lib
from .ops import _stream, gru_cluster_error_tensor


def _storage_span(p):
    """Number of elements of the dense memory block a (possibly permuted-contiguous) parameter occupies."""
    return p.numel()


class FusedAdam(torch.optim.Optimizer):
    TABLE_STEPS = 1 << 20         # per-step Adam scalars kept for the lazy row updates of sparse tables (8 MB)

    def __init__(self, params, lr=5e-4, betas=(0.5, 0.999), eps=1e-8, sparse=()):
        """sparse: embedding tables (subset of params) updated ROW-WISE from compact gradients (ha2g_amd.ops.SparseTable, csrc/sparse.hip):
        they stay out of the flat buffers; results are bit-identical to the dense update (tests/test_gpu_sparse.py)."""
        params = [p for p in params]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        sparse_ids = {id(p) for p in sparse if p.requires_grad}
        ps = [p for p in self.param_groups[0]['params'] if p.requires_grad and id(p) not in sparse_ids]
        assert ps, 'no trainable parameters'
        dev = ps[0].device
        assert all(p.dtype == torch.float32 and p.device == dev for p in ps)
        offs, total = [], 0
        for p in ps:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4                     # keep every tensor 16-byte aligned
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        self.step_t = torch.zeros(1, dtype=torch.int32, device=dev)
        self._views = []
        with torch.no_grad():
            for p, o in zip(ps, offs):
                dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
                src = p.data if dense else p.data.contiguous()
                pv = torch.as_strided(self.flat_p, src.shape, src.stride(), o)
                pv.copy_(src)
                p.data = pv
                gv = torch.as_strided(self.flat_g, src.shape, src.stride(), o)
                p.grad = gv
                self._views.append((p, gv))
        self.total = total
        self.sparse_tables = []
        if sparse_ids:
            from .ops import SparseTable
            self.table = torch.zeros(self.TABLE_STEPS, 2, dtype=torch.float32, device=dev)
            self._host_step = 0
            for p in self.param_groups[0]['params']:
                if id(p) in sparse_ids:
                    st = SparseTable(p, self)
                    p._ha2g_sparse = st
                    self.sparse_tables.append(st)

    def zero_grad(self, set_to_none=False):
        """Gradients live in the flat buffer: zero it in place and keep the views installed."""
        self.flat_g.zero_()
        for p, gv in self._views:
            if p.grad is not gv:
                p.grad = gv
        for st in self.sparse_tables:
            st.pending = []

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        for p, gv in self._views:                                 # a foreign tensor installed as .grad: fold it back
            if p.grad is not gv:
                if p.grad is not None:
                    gv.copy_(p.grad)
                p.grad = gv
        st = _stream()
        # guard = the cluster-GRU error word of this device (None before the first cluster launch): while it is set the kernels below leave the
        # step counter, parameters and moments untouched -- a step whose recurrences timed out never reaches the optimizer state
        guard = gru_cluster_error_tensor(self.flat_p.device)
        gp = guard.data_ptr() if guard is not None else None
        check(lib.ha2g_adam_step_inc_guarded(self.step_t.data_ptr(), gp, st))
        check(lib.ha2g_adam_guarded_f32(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(), self.flat_v.data_ptr(),
                                        self.total, float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']),
                                        self.step_t.data_ptr(), gp, st))
        if self.sparse_tables:
            self._host_step += 1        # informational; past TABLE_STEPS the row kernel recomputes the scalars itself (sparse.hip)
            check(lib.ha2g_adam_scalars(self.step_t.data_ptr(), float(g['lr']), float(g['betas'][0]), float(g['betas'][1]),
                                        self.table.data_ptr(), self.TABLE_STEPS, st))
            for tb in self.sparse_tables:
                tb.step()

    def sync_sparse(self):
        """Replay the pending zero-gradient updates of every row of the sparse tables (call before saving / exporting their weights)."""
        for tb in self.sparse_tables:
            tb.sync()

    def allreduce_grads(self, group=None, async_op=False):
        """Data-parallel: average the flat gradient buffer over ranks (one RCCL all-reduce over xGMI)."""
        from . import ddp
        for tb in self.sparse_tables:
            ddp.exchange_sparse_(tb, group)
        return ddp.average_(self.flat_g, group, async_op)

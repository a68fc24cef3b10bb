import sys, re
sys.path.insert(0, '.')
src = open('tools/gru_fwd3_bench.py').read()
# reuse the setup of the bench tool up to the first timing loop by exec'ing a trimmed copy is brittle; do a small direct bench instead
import torch
from ha2g_amd import ops
from ha2g_amd._lib import check, lib
dev = torch.device('cuda:0'); H, T = 300, 34
def t_us(fn, iters=30):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for B in (384, 128):
    gi = torch.randn(B * T, 6 * H, device=dev) * 0.3
    whh = [torch.randn(3 * H, H, device=dev) * H ** -0.5 for _ in range(2)]
    bh = [torch.randn(3 * H, device=dev) * 0.05 for _ in range(2)]
    st = torch.cuda.current_stream().cuda_stream
    pk3 = torch.empty(2, lib.ha2g_gru_packed3_bytes(), dtype=torch.uint8, device=dev)
    pk3t = torch.empty_like(pk3)
    for d in range(2):
        check(lib.ha2g_gru_pack_whh3(whh[d].data_ptr(), pk3[d].data_ptr(), H, st)); check(lib.ha2g_gru_pack_whh3t(whh[d].data_ptr(), pk3t[d].data_ptr(), H, st))
    y, rs = torch.empty(B, T, 2 * H, device=dev), torch.empty(B, T, 2, 4, H, device=dev)
    xch, err = ops._cluster_scratch(dev)
    dy = torch.randn(B, T, 2 * H, device=dev) * 0.1
    dg, hp = torch.empty(B * T, 8 * H, device=dev), torch.empty(B, T, 2 * H, device=dev)
    f3 = lambda: check(lib.ha2g_gru_layer_fwd_cluster3(gi.data_ptr(), pk3.data_ptr(), bh[0].data_ptr(), bh[1].data_ptr(), y.data_ptr(), rs.data_ptr(), xch.data_ptr(), err.data_ptr(), B, T, H, st))
    b3 = lambda: check(lib.ha2g_gru_layer_bwd_cluster3(dy.data_ptr(), y.data_ptr(), rs.data_ptr(), pk3t.data_ptr(), dg.data_ptr(), hp.data_ptr(), xch.data_ptr(), err.data_ptr(), B, T, H, st))
    for name, fn in (('fwd', f3), ('bwd', b3)):
        r = []
        for dbg in (0, 4, 0, 4):
            lib.ha2g_gru_cluster_debug(dbg); r.append(t_us(fn))
        lib.ha2g_gru_cluster_debug(0)
        print(B, name, 'fast path %.1f %.1f us | write-through only %.1f %.1f us' % (r[0], r[2], r[1], r[3]), 'err', int(err.item()))

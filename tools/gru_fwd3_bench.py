"""The named kernel, before / after: ha2g_gru_layer_fwd_cluster (fp32 MFMA chain) vs ha2g_gru_layer_fwd_cluster3 (three-piece bf16 chain, fp32-class)
at the train step's launch shapes (384 rows = the fused 3-chain pass, 128 rows), with the hand-off ablations of ha2g_gru_cluster_debug.
usage: python tools/gru_fwd3_bench.py"""
import sys
import torch
sys.path.insert(0, '.')
from ha2g_amd import ops
from ha2g_amd._lib import check, lib

dev = torch.device('cuda:0')
H, T = 300, 34


def t_us(fn, iters=20):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


print('%-6s %-28s %10s %10s %10s %12s' % ('rows', 'variant', 'us/launch', 'us/step', 'TFLOP/s', 'of fp32 peak'))
for B in (384, 128):
    gi = torch.randn(B * T, 6 * H, device=dev) * 0.3
    whh = [torch.randn(3 * H, H, device=dev) * H ** -0.5 for _ in range(2)]
    bh = [torch.randn(3 * H, device=dev) * 0.05 for _ in range(2)]
    npk = lib.ha2g_gru_packed_floats(H)
    pk = torch.empty(4, npk, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for d in range(2):
        check(lib.ha2g_gru_pack_whh(whh[d].data_ptr(), pk[d].data_ptr(), pk[2 + d].data_ptr(), H, st))
    pk3 = torch.empty(2, lib.ha2g_gru_packed3_bytes(), dtype=torch.uint8, device=dev)
    for d in range(2):
        check(lib.ha2g_gru_pack_whh3(whh[d].data_ptr(), pk3[d].data_ptr(), H, st))
    y, rs = torch.empty(B, T, 2 * H, device=dev), torch.empty(B, T, 2, 4, H, device=dev)
    y3 = torch.empty_like(y)
    xch, err = ops._cluster_scratch(dev)
    f32 = lambda: check(lib.ha2g_gru_layer_fwd_cluster(gi.data_ptr(), pk.data_ptr(), bh[0].data_ptr(), bh[1].data_ptr(), y.data_ptr(), rs.data_ptr(), xch.data_ptr(), err.data_ptr(), B, T, H, st))
    f3 = lambda: check(lib.ha2g_gru_layer_fwd_cluster3(gi.data_ptr(), pk3.data_ptr(), bh[0].data_ptr(), bh[1].data_ptr(), y3.data_ptr(), rs.data_ptr(), xch.data_ptr(), err.data_ptr(), B, T, H, st))
    flops = 2.0 * B * T * 2 * 3 * H * H + 12.0 * B * T * 2 * H
    for name, fn in (('fp32 chain (rounds 1-3)', f32), ('three-piece bf16 chain', f3)):
        for dbg, dn in ((0, ''), (2, ' [no exchange]'), (1, ' [no poll wait]')):
            lib.ha2g_gru_cluster_debug(dbg)
            us = t_us(fn)
            print('%-6d %-28s %10.1f %10.2f %10.1f %12.3f' % (B, name + dn, us, us / T, flops / us / 1e6, flops / us / 1e6 / 157.3))
        lib.ha2g_gru_cluster_debug(0)
    f32(); f3()
    torch.cuda.synchronize()
    print('       max |y3 - y32| / max |y32| = %.3e; err word %d' % (float((y3 - y).abs().max() / y.abs().max()), int(err.item())))
    # ---- BPTT twin on the same rows ----
    dy = torch.randn(B, T, 2 * H, device=dev) * 0.1
    dg, hp, dg3 = torch.empty(B * T, 8 * H, device=dev), torch.empty(B, T, 2 * H, device=dev), torch.empty(B * T, 8 * H, device=dev)
    pk3t = torch.empty(2, lib.ha2g_gru_packed3_bytes(), dtype=torch.uint8, device=dev)
    for d in range(2):
        check(lib.ha2g_gru_pack_whh3t(whh[d].data_ptr(), pk3t[d].data_ptr(), H, st))
    lib.ha2g_gemm_set_mode(0)
    b32 = lambda: check(lib.ha2g_gru_layer_bwd_cluster(dy.data_ptr(), y.data_ptr(), rs.data_ptr(), pk[2].data_ptr(), dg.data_ptr(), hp.data_ptr(), xch.data_ptr(), err.data_ptr(), B, T, H, st))
    b3 = lambda: check(lib.ha2g_gru_layer_bwd_cluster3(dy.data_ptr(), y.data_ptr(), rs.data_ptr(), pk3t.data_ptr(), dg3.data_ptr(), hp.data_ptr(), xch.data_ptr(), err.data_ptr(), B, T, H, st))
    for name, fn in (('BPTT fp32 chain', b32), ('BPTT three-piece chain', b3)):
        abl = ((0, ''),) if fn is b32 else ((0, ''), (2, ' [no exchange]'), (1, ' [no poll wait]'), (8, ' [no split]'), (16, ' [no MFMA]'), (32, ' [no dg stores]'),
                                            (64, ' [no operand loads]'), (96, ' [no HBM traffic]'), (2 + 8 + 16 + 96, ' [phase 1 + barriers only]'))
        for dbg, dn in abl:
            lib.ha2g_gru_cluster_debug(dbg)
            us = t_us(fn)
            print('%-6d %-28s %10.1f %10.2f %10.1f %12.3f' % (B, name + dn, us, us / T, flops / us / 1e6, flops / us / 1e6 / 157.3))
        lib.ha2g_gru_cluster_debug(0)
    # where a BPTT step goes: shader-clock cycles between eight points of the step loop, wave 0 of the five members of cluster 0 (debug bit 7)
    import numpy as np
    names = ['gate gradients + dg stores', 'split3 + LDS writes', 'barrier A', 'fragment reads + 4 foreign k tiles', 'own k tile + gather issue', 'barrier B',
             'poll wait + carry', 'operand hand-over + loads']
    for dbg, dn in ((128, ''), (128 + 2, ' [no exchange]')):
        lib.ha2g_gru_cluster_debug(dbg)
        us = t_us(b3, iters=5)
        tab = np.zeros(5 * 16, np.uint64)
        check(lib.ha2g_gru_cluster_prof(tab.ctypes.data))
        tab = tab.reshape(5, 16).astype(np.float64)
        steps = tab[0, 8]
        tot = tab[:, :8].sum(1) / steps
        print('%-6d BPTT step probe%s: %.1f us / launch (probed), cycles per step and member (0..4), mean:' % (B, dn, us))
        for k, nm in enumerate(names):
            print('         %-36s %s   %7.0f' % (nm, ' '.join('%6.0f' % (tab[m, k] / steps) for m in range(5)), tab[:, k].mean() / steps))
        print('         %-36s %s   %7.0f  (= %.2f us per step at %.2f GHz if the probed launch took its time in the loop)' % (
            'sum', ' '.join('%6.0f' % v for v in tot), tot.mean(), us / T, tot.mean() / (us / T) / 1e3))
    lib.ha2g_gru_cluster_debug(0)
    # the same launch beside a stream of weight-gradient-shaped products on a second queue (what the train step's side queue does to it)
    ga, gb = torch.randn(B * T, 900, device=dev), torch.randn(B * T, 600, device=dev)
    gout = torch.empty(900, 600, device=dev)
    s2 = torch.cuda.Stream(dev)
    def contended(fn, iters=10):
        fn(); torch.cuda.synchronize()
        with torch.cuda.stream(s2):
            for _ in range(iters * 6):
                ops.gemm(ga, gb, transa=True, out=gout)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / iters * 1e3
    for dbg, dn in ((0, ''), (2, ' [no exchange]'), (32, ' [no dg stores]'), (64, ' [no operand loads]'), (96, ' [no HBM traffic]')):
        lib.ha2g_gru_cluster_debug(dbg)
        us = contended(b3)
        print('%-6d %-28s %10.1f %10.2f   beside dW GEMMs on a second queue' % (B, 'BPTT three-piece' + dn, us, us / T))
    lib.ha2g_gru_cluster_debug(0)
    us = contended(f3)
    print('%-6d %-28s %10.1f %10.2f   beside dW GEMMs on a second queue' % (B, 'forward three-piece', us, us / T))
    b32(); b3()
    torch.cuda.synchronize()
    print('       max |dg3 - dg32| / max |dg32| = %.3e; err word %d' % (float((dg3 - dg).abs().max() / dg.abs().max()), int(err.item())))
    from ha2g_amd._lib import DEFAULT_GEMM_MODE
    lib.ha2g_gemm_set_mode(DEFAULT_GEMM_MODE)

"""Timing ablations (ha2g_conv_planes_debug bits) of the patch-resident plane kernel pconv_r_kernel and the q kernel beside it, data gradient at the trunk
shapes, B = 128: full / no DMA after the prologue / no MFMA / neither / no k loop / no k loop + no stores / no stores / empty launch.
usage: python tools/r_kernel_ablate.py"""
import sys
import torch
sys.path.insert(0, '.')
from ha2g_amd import ops, wav_engine as we
from ha2g_amd._lib import lib

dev = torch.device('cuda:0')
B = 128


def t_us(fn, iters=30):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


print('%-26s %8s %8s %8s %8s %9s %12s %9s %8s' % ('shape / kernel', 'full', 'no DMA', 'no MFMA', 'neither', 'no k loop', 'no k, no st', 'no stores', 'empty') + '  no setprio     full')
for H, W, C in ((64, 35, 64), (32, 18, 128), (16, 9, 256)):
    dy = torch.randn(B, H, W, C, device=dev)
    w = torch.randn(C, 3, 3, C, device=dev) * 0.05
    out = torch.empty(B, H, W, C, device=dev)
    pl = ops.to_planes(dy, 3)
    wpl = we.weight_planes(w, 3)
    fn = lambda: lib.ha2g_conv2d_dgrad_planes_np_f32(pl.data_ptr(), pl.stride(0), wpl.data_ptr(), wpl.stride(0), 3, out.data_ptr(), B, H, W, C, C, 3, 3, 1, 1, 0.0,
                                                     torch.cuda.current_stream().cuda_stream)
    for name, sw in (('q', 8), ('r', 0)):
        lib.ha2g_conv_planes_tile3(sw)
        ts = []
        for bits in (0, 1, 2, 3, 4, 12, 8, 32, 64, 0):
            lib.ha2g_conv_planes_debug(bits)
            ts.append(t_us(fn))
        lib.ha2g_conv_planes_debug(0)
        print('C=%-3d %2dx%-2d %s kernel       ' % (C, H, W, name) + ' '.join('%8.1f' % t for t in ts))
    lib.ha2g_conv_planes_tile3(0)

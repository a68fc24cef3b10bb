"""The patch-resident plane kernel (pconv_r_kernel) against the q kernel on the trunk's three 3x3 / stride-1 shapes at B = 128, forward and data
gradient, alone.  usage: python tools/r_kernel_bench.py"""
import sys
import torch
sys.path.insert(0, '.')
from ha2g_amd import ops, wav_engine as we
from ha2g_amd._lib import lib
dev = 'cuda:0'
B = 128


def t_us(fn, iters=40):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


print('%-34s %10s %10s %8s   TFLOP/s (fp32-eq.)  of 417' % ('shape', 'q kernel', 'r kernel', 'ratio'))
for C, H, W in ((64, 64, 35), (128, 32, 18), (256, 16, 9)):
    x = torch.randn(B, H, W, C, device=dev)
    w = torch.randn(C, 3, 3, C, device=dev) * 0.05
    xp, wp = ops.to_planes(x, 3), ops.to_planes(w.contiguous(), 3)
    wpl = we.weight_planes(w, 3)
    out = torch.empty(B, H, W, C, device=dev)
    flops = 2.0 * B * H * W * C * 9 * C
    for name, fn in (('fwd', lambda: we.conv_fwd_planes(xp, wp, x.shape, 1, 1, we.ACT_RELU)),
                     ('dgrad', lambda: we.conv_dgrad_planes(xp, w, (B, H, W, C), 1, 1, out=out, beta=0.0))):
        res = {}
        for rep in range(2):
            for k, sw in (('q', 8), ('r', 0)):
                lib.ha2g_conv_planes_tile3(sw)
                res[k] = min(res.get(k, 1e9), t_us(fn))
        lib.ha2g_conv_planes_tile3(0)
        tf = flops / (res['r'] * 1e-6) / 1e12
        print('%-34s %8.1f us %8.1f us %8.2f   %8.1f          %.2f' % ('C=%d %dx%d %s' % (C, H, W, name), res['q'], res['r'], res['q'] / res['r'], tf, tf / (2500.0 / 6)))

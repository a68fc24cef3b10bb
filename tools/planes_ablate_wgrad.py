"""Timing ablation of the three-piece plane-based weight-gradient kernel (+ its wide reduce): full vs no DMA after the first tile (ha2g_conv_planes_debug(1)); B = 128."""
import sys
import torch
sys.path.insert(0, '.')
from ha2g_amd import ops, wav_engine as we
from ha2g_amd._lib import lib

dev = torch.device('cuda:0')


def t_us(fn, iters=10):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


B = 128
print('%-24s %9s %9s' % ('shape', 'full', 'no DMA'))
for H, W, C in ((64, 35, 64), (32, 18, 128), (16, 9, 256)):
    x = torch.randn(B, H, W, C, device=dev); dy = torch.randn(B, H, W, C, device=dev)
    w = torch.zeros(C, 3, 3, C, device=dev)
    xp, dp = ops.to_planes(x, 3), ops.to_planes(dy, 3)
    fn = lambda: we.conv_wgrad_planes(xp, dp, w, x.shape)
    ts = []
    for bits in (0, 1):
        lib.ha2g_conv_planes_debug(bits)
        ts.append(t_us(fn))
    lib.ha2g_conv_planes_debug(0)
    print('wgrad C=%-3d %3dx%-3d      %9.1f %9.1f' % (C, H, W, ts[0], ts[1]))

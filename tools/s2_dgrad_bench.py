"""Stride-2 data gradients (first block of trunk layers 2-4: 3x3 and 1x1) on three-piece planes: q kernel with the parity classes over grid.z (default) vs the
32x32 kernels (ha2g_conv_planes_tile3(7)).  usage: python tools/s2_dgrad_bench.py"""
import sys
import torch
sys.path.insert(0, '.')
from ha2g_amd import ops, wav_engine as we
from ha2g_amd._lib import lib
dev = torch.device('cuda:0')


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


B = 128
print('%-40s %12s %12s' % ('shape (input of the conv)', 'q kernel us', '32x32 us'))
for (Cin, Cout, k, H, W) in ((32, 64, 3, 128, 70), (32, 64, 1, 128, 70), (64, 128, 3, 64, 35), (64, 128, 1, 64, 35), (128, 256, 3, 32, 18), (128, 256, 1, 32, 18)):
    pad = 1 if k == 3 else 0
    OH, OW = (H + 2 * pad - k) // 2 + 1, (W + 2 * pad - k) // 2 + 1
    dy = torch.randn(B, OH, OW, Cout, device=dev)
    w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    dyp = ops.to_planes(dy)
    out = torch.zeros(B, H, W, Cin, device=dev)
    fn = lambda: we.conv_dgrad_planes(dyp, w, (B, H, W, Cin), 2, pad, out=out, beta=0.0)
    r = []
    for t in (0, 7):
        lib.ha2g_conv_planes_tile3(t)
        r.append(t_us(fn))
    lib.ha2g_conv_planes_tile3(0)
    print('Cin %3d Cout %3d k %d  %3dx%-3d                %12.1f %12.1f' % (Cin, Cout, k, H, W, r[0], r[1]))

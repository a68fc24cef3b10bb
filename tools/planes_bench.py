"""A/B of the plane-based data gradient (csrc/conv_planes.hip) against the round-2 split implicit GEMM, per trunk shape at B = 128, in one
process, interleaved rounds (guide 5.4 rule 24).  usage: python tools/planes_bench.py [rounds]"""
import sys
import torch
sys.path.insert(0, '.')
from ha2g_amd import ops, wav_engine as we
from ha2g_amd._lib import lib

dev = torch.device('cuda:0')
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = 128


def t_us(fn, iters=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


print('%-28s %10s %10s %8s %10s %8s' % ('shape', 'old us', 'planes us', 'ratio', 'TF(f32eq)', 'of 833'))
for H, W, C in ((64, 35, 64), (32, 18, 128), (16, 9, 256)):
    dy = torch.randn(B, H, W, C, device=dev)
    w = torch.randn(C, 3, 3, C, device=dev) * 0.05
    pl = ops.to_planes(dy)
    shp = (B, H, W, C)
    old = lambda: we.conv_dgrad(dy, w, shp, 1, 1)
    new = lambda: we.conv_dgrad_planes(pl, w, shp, 1, 1)
    for f in (old, new):
        f(); f()
    torch.cuda.synchronize()
    to, tn = [], []
    for _ in range(rounds):
        to.append(t_us(old)); tn.append(t_us(new))
    to.sort(); tn.sort()
    mo, mn = to[len(to) // 2], tn[len(tn) // 2]
    fl = 2.0 * B * H * W * C * C * 9
    print('dgrad C=%-3d %3dx%-3d          %10.1f %10.1f %8.2f %10.1f %8.3f' % (C, H, W, mo, mn, mo / mn, fl / mn / 1e6, fl / mn / 1e6 / 833.0))
    ref = new().clone()
    for depth in (2, 3, 4):                                  # LDS ring depth of the plane kernel (ha2g_conv_planes_ring); 0 = default
        lib.ha2g_conv_planes_ring(depth)
        assert torch.equal(new(), ref), depth
        new(); torch.cuda.synchronize()
        ts_ = sorted(t_us(new) for _ in range(rounds))
        print('      ring depth %d: %8.1f us' % (depth, ts_[len(ts_) // 2]))
    lib.ha2g_conv_planes_ring(0)
    lib.ha2g_conv_planes_waves(8)                            # eight-wave workgroups, twice the tile
    assert torch.equal(new(), ref), 'waves 8'
    new(); torch.cuda.synchronize()
    ts_ = sorted(t_us(new) for _ in range(rounds))
    print('      8 waves     : %8.1f us' % ts_[len(ts_) // 2])
    lib.ha2g_conv_planes_waves(4)
print()
print('%-28s %10s %10s %8s %10s %8s' % ('shape', 'old us', 'planes us', 'ratio', 'TF(f32eq)', 'of 833'))
for H, W, C in ((64, 35, 64), (32, 18, 128), (16, 9, 256)):
    x = torch.randn(B, H, W, C, device=dev)
    dy = torch.randn(B, H, W, C, device=dev)
    w = torch.zeros(C, 3, 3, C, device=dev)
    xp, dp = ops.to_planes(x), ops.to_planes(dy)
    old = lambda: we.conv_wgrad(x, dy, w, 1, 1)
    new = lambda: we.conv_wgrad_planes(xp, dp, w, x.shape)
    new_split = lambda: we.conv_wgrad_planes(ops.to_planes(x), dp, w, x.shape)      # including the on-demand split of x
    for f in (old, new, new_split):
        f(); f()
    torch.cuda.synchronize()
    to, tn, ts = [], [], []
    for _ in range(rounds):
        to.append(t_us(old)); tn.append(t_us(new)); ts.append(t_us(new_split))
    to.sort(); tn.sort(); ts.sort()
    mo, mn, ms = to[len(to) // 2], tn[len(tn) // 2], ts[len(ts) // 2]
    fl = 2.0 * B * H * W * C * C * 9
    print('wgrad C=%-3d %3dx%-3d          %10.1f %10.1f %8.2f %10.1f %8.3f   (+ split of x: %.1f us)' % (C, H, W, mo, mn, mo / mn, fl / mn / 1e6, fl / mn / 1e6 / 833.0, ms))

"""Tile-shape sweep of the plane-based split-bf16 conv kernel (mode bit 0 forces it on the forward conv = same kernel the data
gradient uses) and of the packed-word split kernel on wgrad: which tile config is best when the inner product is 2.7x cheaper."""
import sys, torch
sys.path.insert(0, '.')
from ha2g_amd import wav_engine as we
from ha2g_amd._lib import lib
dev = torch.device('cuda:0')
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
B = 128
names = {-1: 'auto', 0: '256x32', 1: '128x64', 2: '128x128', 3: '64x64', 4: '64x128'}
for H, W, C in ((64, 35, 64), (32, 18, 128), (16, 9, 256)):
    x = torch.randn(B, H, W, C, device=dev).clamp_min(0); w = torch.randn(C, 3, 3, C, device=dev) * 0.05
    dy = torch.randn(B, H, W, C, device=dev)
    fl = 2.0 * B * H * W * C * C * 9
    row = []
    for cfg in (-1, 1, 2, 3, 4):
        lib.ha2g_conv_debug_cfg(cfg)
        t = timeit(lambda: we.conv_dgrad(dy, w, (B, H, W, C), 1, 1))
        row.append('%s %.0fus %.0fTF' % (names[cfg], t, fl / t / 1e6))
    lib.ha2g_conv_debug_cfg(-1)
    print('dgrad C=%d: ' % C + ' | '.join(row))

"""Time the audio encoder's four dominant 3x3 convolutions (forward / data-gradient / weight-gradient) per tile shape."""
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
import os
if os.environ.get('BK16'): lib.ha2g_conv_debug_cfg(100)
shapes = [(128, 70, 32), (64, 35, 64), (32, 18, 128), (16, 9, 256)]
names = ['256x32', '128x64', '128x128', '64x64', '64x128']
for H, W, C in shapes:
    x = torch.randn(B, H, W, C, device=dev); w = torch.randn(C, 3, 3, C, device=dev) * 0.05; dy = torch.randn(B, H, W, C, device=dev)
    fl = 2.0 * B * H * W * C * C * 9
    row = []
    for cfg in range(5):
        if cfg != 0 and C == 32 and names[cfg].endswith(('x64', 'x128')) and False: continue
        lib.ha2g_conv_debug_cfg(cfg)
        f = timeit(lambda: we.conv_fwd(x, w, None, 1, 1, 1)); d = timeit(lambda: we.conv_dgrad(dy, w, x.shape, 1, 1))
        row.append('%s fwd %.0fus %.0fTF dgrad %.0fus' % (names[cfg], f, fl / f / 1e6, d))
    lib.ha2g_conv_debug_cfg(-1)
    f = timeit(lambda: we.conv_fwd(x, w, None, 1, 1, 1)); g = timeit(lambda: we.conv_wgrad(x, dy, w, 1, 1))
    print('C=%d %dx%d: ' % (C, H, W) + ' | '.join(row) + ' || heuristic fwd %.0fus, wgrad %.0fus %.0fTF' % (f, g, fl / g / 1e6))

"""wgrad block-count sweep with the split-bf16 inner product (debug knob 10000+n = target workgroups per launch)."""
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
for H, W, C in ((128, 70, 32), (64, 35, 64), (32, 18, 128), (16, 9, 256)):
    x = torch.randn(B, H, W, C, device=dev).clamp_min(0); w = torch.randn(C, 3, 3, C, device=dev) * 0.05
    dy = torch.randn(B, H, W, C, device=dev)
    fl = 2.0 * B * H * W * C * C * 9
    row = []
    for nb in (0, 256, 384, 512, 768, 1024, 1536, 2048):
        lib.ha2g_conv_debug_cfg(10000 + nb)
        try:
            t = timeit(lambda: we.conv_wgrad(x, dy, w, 1, 1))
        except AssertionError:
            t = float("nan")
        row.append('%d: %.0fus' % (nb, t))
    lib.ha2g_conv_debug_cfg(10000)
    print('wgrad C=%d (%.0f TF at 1024): ' % (C, 0) + ' | '.join(row))

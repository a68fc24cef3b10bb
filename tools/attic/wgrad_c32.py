"""Weight gradient of the 32-channel 3x3 convolutions (layer1, 128 x 70 maps, B = 128): narrow (3 x 128-column tiles) vs wide (one 384-column tile)
implicit-GEMM configuration x workgroup target."""
import sys, torch
sys.path.insert(0, '.')
from ha2g_amd import wav_engine as we
from ha2g_amd._lib import lib
dev = torch.device('cuda:0')
x = torch.randn(128, 128, 70, 32, device=dev); dy = torch.randn(128, 128, 70, 32, device=dev)
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
ref = None
W0 = torch.empty(32, 3, 3, 32, device=dev)
for wide in (0, 1):
    for blocks in (0, 256, 512, 768, 1024, 1536):
        lib.ha2g_conv_debug_cfg(20000 + wide); lib.ha2g_conv_debug_cfg(10000 + blocks)
        out = we.conv_wgrad(x, dy, W0, 1, 1)
        if ref is None: ref = out.clone()
        err = float((out - ref).abs().max() / ref.abs().max())
        print('wide %d blocks %4d: %.0f us  (rel diff vs first %.1e)' % (wide, blocks, timeit(lambda: we.conv_wgrad(x, dy, W0, 1, 1)), err))
lib.ha2g_conv_debug_cfg(20001); lib.ha2g_conv_debug_cfg(10000)

"""Data gradient of the 32->32 channel 3x3 convolution: split-bf16 direct kernel vs split-bf16 implicit GEMM vs fp32 direct."""
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
for B, H, W, C in ((128, 128, 70, 32),):
    w = torch.randn(C, 3, 3, C, device=dev) * 0.05
    dy = torch.randn(B, H, W, C, device=dev)
    for mode, name in ((0, 'split-bf16 direct'), (4, 'split-bf16 implicit GEMM')) + (((2, 'fp32 direct'),) if C == 32 else ()):
        lib.ha2g_conv_debug_direct_c32(mode)
        print('C=%d %-26s %.0f us' % (C, name, timeit(lambda: we.conv_dgrad(dy, w, (B, H, W, C), 1, 1))))
    lib.ha2g_conv_debug_direct_c32(0)

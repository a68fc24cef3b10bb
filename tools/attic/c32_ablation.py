import sys, torch
sys.path.insert(0, '.')
from ha2g_amd import wav_engine as we
from ha2g_amd._lib import lib
dev = torch.device('cuda:0')
x = torch.randn(128, 128, 70, 32, device=dev); w = torch.randn(32, 3, 3, 32, device=dev) * 0.05
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for dbg, name in ((1, 'full'), (1 | 0x10, 'no epilogue stores'), (1 | 0x20, 'no next-patch fetch'), (1 | 0x30, 'neither')):
    lib.ha2g_conv_debug_direct_c32(dbg)
    print(name, '%.0f us' % timeit(lambda: we.conv_fwd(x, w, None, 1, 1, 0)))
xz = torch.zeros_like(x)
lib.ha2g_conv_debug_direct_c32(1)
print('zeros input', '%.0f us' % timeit(lambda: we.conv_fwd(xz, w, None, 1, 1, 0)))
lib.ha2g_conv_debug_direct_c32(0)
print('implicit GEMM', '%.0f us' % timeit(lambda: we.conv_fwd(x, w, None, 1, 1, 0)))

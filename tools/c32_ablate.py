"""Timing ablations of the prefetching 32-channel three-piece direct convolution (conv3x3_x3p_kernel, data gradient shape at B = 128, 128x70):
full / no MFMA / no LDS fill / no stores / no loads / combinations.  usage: python tools/c32_ablate.py"""
import sys
import torch
sys.path.insert(0, '.')
from ha2g_amd._lib import lib, check
dev = torch.device('cuda:0')
B, H, W, C = 128, 128, 70, 32
dy = torch.randn(B, H, W, C, device=dev)
w = (torch.randn(C, 3, 3, C, device=dev) * 0.05)
wt = w.permute(3, 1, 2, 0).contiguous()
dx = torch.empty(B, H, W, C, device=dev)
st = torch.cuda.current_stream().cuda_stream


def t_us(iters=60):
    fn = lambda: check(lib.ha2g_conv2d_dgrad_f32(dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), B, H, W, C, C, 3, 3, 1, 1, 0.0, st))
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for label, form, bits in (('first form (single buffer, conv3x3_x3_kernel<32,3>)', 'x3', 0), ('prefetching form (conv3x3_x3p_kernel)', 'x3p', 0),
                          ('anti-phase form (conv3x3_c32pp_kernel)', 'pp', 0), ('  no MFMA', 'pp', 1), ('  no LDS fill (split + writes)', 'pp', 2), ('  no stores', 'pp', 4),
                          ('  no loads', 'pp', 8), ('  no MFMA, no stores', 'pp', 5), ('  nothing but the barriers', 'pp', 15), ('  only MFMA (no fill, stores, loads)', 'pp', 14), ('anti-phase form, groups by wave parity instead of wave >> 2', 'pp', 256), ('  only MFMA', 'pp', 256 | 14)):
    lib.ha2g_conv_c32_prefetch({'x3': 32, 'x3p': 33, 'pp': 1}[form] | ((bits & 15) << 1) | (256 if bits & 256 else 0))
    print('%-52s %8.1f us' % (label, t_us()))
for label, sw in (('anti-phase form again (after the ablations: clocks settled)', 1), ('anti-phase form again', 1)):
    lib.ha2g_conv_c32_prefetch(sw)
    print('%-52s %8.1f us' % (label, t_us()))
lib.ha2g_conv_c32_prefetch(1)

"""Timing ablations of the plane-based data-gradient kernel (ha2g_conv_planes_debug): full / no DMA / no MFMA / neither, at the trunk shapes
(B = 128) and at 4x the pixels (tile quantisation removed), for the two-piece (mode 6) and the three-piece (mode 70, default) kernels, every
tile shape of the three-piece kernel and both LDS ring depths.  usage: python tools/planes_ablate.py"""
import sys
import torch
sys.path.insert(0, '.')
from ha2g_amd import ops, wav_engine as we
from ha2g_amd._lib import DEFAULT_GEMM_MODE, lib

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


print('%-44s %9s %9s %9s %9s   %s   %s' % ('shape', 'full', 'no DMA', 'no MFMA', 'neither', 'TF(f32eq) full', '[q kernel: no k loop | no k loop, no stores | no stores | reads before DMA | empty launch]'))
for B in (128, 512):
    for H, W, C in ((64, 35, 64), (32, 18, 128), (16, 9, 256)):
        dy = torch.randn(B, H, W, C, device=dev)
        w = torch.randn(C, 3, 3, C, device=dev) * 0.05
        out = torch.empty(B, H, W, C, device=dev)
        for mode, tiles in ((6, (0,)), (70, (1, 3, 4, 5))):
            lib.ha2g_gemm_set_mode(mode)
            pl = ops.to_planes(dy)
            wpl = we.weight_planes(w, pl.shape[0])
            fn = lambda: lib.ha2g_conv2d_dgrad_planes_np_f32(pl.data_ptr(), pl.stride(0), wpl.data_ptr(), wpl.stride(0), pl.shape[0], out.data_ptr(), B, H, W,
                                                             C, C, 3, 3, 1, 1, 0.0, torch.cuda.current_stream().cuda_stream)
            for tile in tiles:
                lib.ha2g_conv_planes_tile3(tile)
                for ring in ((2,) if tile >= 4 else (2, 3)):      # LDS ring depth (ha2g_conv_planes_ring; the ping-pong kernel has none)
                    lib.ha2g_conv_planes_ring(ring)
                    ts = []
                    for bits in (0, 1, 2, 3):
                        lib.ha2g_conv_planes_debug(bits)
                        ts.append(t_us(fn))
                    extra = ''
                    if tile == 5:                         # q kernel: per-launch fixed costs (bit 2 = skip the k loop, bit 3 = skip the output stores)
                        ex = []
                        for bits in (4, 12, 8, 16, 32):
                            lib.ha2g_conv_planes_debug(bits)
                            ex.append(t_us(fn))
                        extra = '   [%.1f | %.1f | %.1f | %.1f | %.1f]' % tuple(ex)
                    lib.ha2g_conv_planes_debug(0)
                    fl = 2.0 * B * H * W * C * C * 9
                    print('B=%-3d C=%-3d %3dx%-3d np %d tile %d ring %d        %9.1f %9.1f %9.1f %9.1f   %.1f%s' % (
                        B, C, H, W, pl.shape[0], tile, ring, ts[0], ts[1], ts[2], ts[3], fl / ts[0] / 1e6, extra))
            lib.ha2g_conv_planes_ring(0)
            lib.ha2g_conv_planes_tile3(0)
lib.ha2g_gemm_set_mode(DEFAULT_GEMM_MODE)

"""BatchNorm / SE streaming passes of the audio tower in isolation (B = 128), per trunk shape: microseconds and algorithmic GB/s.
usage: python tools/bn_bench.py"""
import sys
import torch
sys.path.insert(0, '.')
from ha2g_amd import ops
from ha2g_amd._lib import lib, check
from ha2g_amd.ops import _stream, workspace

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
print('%-14s %-34s %9s %9s' % ('shape', 'pass', 'us', 'GB/s'))
for H, W, C in ((128, 70, 32), (64, 35, 64), (32, 18, 128), (16, 9, 256)):
    rows = B * H * W
    x = torch.randn(rows, C, device=dev)
    dy = torch.randn(rows, C, device=dev)
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    mean, invstd = ops.bn_stats(x, None, None, 0.1, 1e-5)
    nb = 4.0 * rows * C
    y = torch.empty_like(x)
    tests = [
        ('bn_stats (1R)', lambda: ops.bn_stats(x, None, None, 0.1, 1e-5), nb),
        ('bn_apply (1R 1W)', lambda: ops.bn_apply(x, mean, invstd, gamma, beta, out=y), 2 * nb),
        ('bn_bwd stats+apply (4R 1W)', lambda: ops.bn_bwd(dy, x, mean, invstd, gamma), 5 * nb),
        ('bn_bwd planes only (4R 1W)', lambda: ops.bn_bwd(dy, x, mean, invstd, gamma, need_dx=False, planes=True), 5 * nb),
        ('bn_bwd stats only (2R)', lambda: ops.bn_bwd(dy, x, mean, invstd, gamma, need_dx=False), 2 * nb),
        ('eltwise add (2R 1W)', lambda: ops.eltwise(ops.OP_ADD, x, dy, out=y), 3 * nb),
    ]
    x4 = x.view(B, H * W, C)
    s = torch.rand(B, C, device=dev)
    out = torch.empty_like(x4)
    tests.append(('se_scale_add_relu (2R 1W)', lambda: check(lib.ha2g_se_scale_add_relu_f32(x4.data_ptr(), s.data_ptr(), dy.data_ptr(), out.data_ptr(), B, H * W, C, _stream())), 3 * nb))
    dres, db2 = torch.empty_like(x4), torch.empty_like(x4)
    dpool = torch.rand(B, C, device=dev)
    tests.append(('se_bwd_apply (2R 2W)', lambda: check(lib.ha2g_se_bwd_apply_f32(dy.data_ptr(), x4.data_ptr(), s.data_ptr(), dpool.data_ptr(), dres.data_ptr(), db2.data_ptr(), B, H * W, C, _stream())), 4 * nb))
    ds = torch.empty(B, C, device=dev)
    tests.append(('se_bwd_scale (3R)', lambda: check(lib.ha2g_se_bwd_scale_f32(dy.data_ptr(), x4.data_ptr(), y.data_ptr(), ds.data_ptr(), B, H * W, C, s.data_ptr(), workspace(dev).data_ptr(), _stream())), 3 * nb))
    for name, fn, nbytes in tests:
        us = t_us(fn)
        print('C=%-3d %3dx%-3d  %-34s %9.1f %9.0f' % (C, H, W, name, us, nbytes / us / 1e3))

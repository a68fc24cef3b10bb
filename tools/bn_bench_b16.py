"""BatchNorm / SE streaming passes over fp32 and bf16 tensors side by side (bf16-storage mode, BASELINE config 5), per trunk shape at a given
batch: microseconds and algorithmic GB/s.   usage: python tools/bn_bench_b16.py [B=256]"""
import sys
import torch
sys.path.insert(0, '.')
from ha2g_amd import ops, wav_b16 as wb, wav_engine as we
from ha2g_amd._lib import lib, check
from ha2g_amd.ops import _stream, workspace

dev = torch.device('cuda:0')
BF = torch.bfloat16


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


class Sink:
    def __init__(self, P):
        self.P, self.G = P, {}

    @staticmethod
    def tgt(t):
        return None


B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
print('%-14s %-30s %9s %9s %9s %9s' % ('shape', 'pass', 'f32 us', 'GB/s', 'bf16 us', 'GB/s'))
for H, W, C in ((128, 70, 32), (64, 35, 64), (32, 18, 128), (16, 9, 256)):
    rows = B * H * W
    x = torch.randn(rows, C, device=dev)
    dy = torch.randn(rows, C, device=dev)
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    mean, invstd = ops.bn_stats(x, None, None, 0.1, 1e-5)
    x16, dy16 = x.to(BF).view(B, H, W, C), dy.to(BF).view(B, H, W, C)
    bn = we._BN(gamma, beta, None, None, None)
    sink = Sink({'bn': bn})
    y = torch.empty_like(x)
    y16 = torch.empty_like(x16)
    nb = 4.0 * rows * C
    ws = workspace(dev)
    m2, i2 = torch.empty_like(mean), torch.empty_like(invstd)
    x4 = x.view(B, H * W, C)
    s = torch.rand(B, C, device=dev)
    out, out16 = torch.empty_like(x4), torch.empty_like(x16)
    dres, db2 = torch.empty_like(x4), torch.empty_like(x4)
    dres16, db216 = torch.empty_like(x16), torch.empty_like(x16)
    dpool = torch.rand(B, C, device=dev)
    ds = torch.empty(B, C, device=dev)
    pooled = torch.empty(B, C, device=dev)
    tests = [
        ('bn_stats (1R)', nb,
         lambda: ops.bn_stats(x, None, None, 0.1, 1e-5),
         lambda: check(lib.ha2g_bn_stats_b16(x16.data_ptr(), rows, C, m2.data_ptr(), i2.data_ptr(), 0, 0, 0.1, 1e-5, ws.data_ptr(), _stream()))),
        ('bn_apply (1R 1W)', 2 * nb,
         lambda: ops.bn_apply(x, mean, invstd, gamma, beta, out=y),
         lambda: check(lib.ha2g_bn_apply_b16(x16.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y16.data_ptr(), rows, C, 0, _stream()))),
        ('bn_apply_pool (1R 1W)', 2 * nb,
         lambda: check(lib.ha2g_bn_apply_pool_f32(x.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), B, H * W, C, pooled.data_ptr(), ws.data_ptr(), _stream())),
         lambda: check(lib.ha2g_bn_apply_pool_b16(x16.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y16.data_ptr(), B, H * W, C, pooled.data_ptr(), ws.data_ptr(), _stream()))),
        ('bn_bwd stats+apply (4R 1W)', 5 * nb,
         lambda: ops.bn_bwd(dy, x, mean, invstd, gamma),
         lambda: wb.gbn(sink, 'bn', dy16, x16, mean, invstd)),
        ('se_scale_add_relu (2R 1W)', 3 * nb,
         lambda: check(lib.ha2g_se_scale_add_relu_f32(x4.data_ptr(), s.data_ptr(), dy.data_ptr(), out.data_ptr(), B, H * W, C, _stream())),
         lambda: check(lib.ha2g_se_scale_add_relu_b16(x16.data_ptr(), s.data_ptr(), dy16.data_ptr(), out16.data_ptr(), B, H * W, C, _stream()))),
        ('se_bwd_apply (2R 2W)', 4 * nb,
         lambda: check(lib.ha2g_se_bwd_apply_f32(dy.data_ptr(), x4.data_ptr(), s.data_ptr(), dpool.data_ptr(), dres.data_ptr(), db2.data_ptr(), B, H * W, C, _stream())),
         lambda: check(lib.ha2g_se_bwd_apply_b16(dy16.data_ptr(), x16.data_ptr(), s.data_ptr(), dpool.data_ptr(), dres16.data_ptr(), db216.data_ptr(), B, H * W, C, _stream()))),
        ('se_bwd_scale (3R)', 3 * nb,
         lambda: check(lib.ha2g_se_bwd_scale_f32(dy.data_ptr(), x4.data_ptr(), y.data_ptr(), ds.data_ptr(), B, H * W, C, s.data_ptr(), ws.data_ptr(), _stream())),
         lambda: check(lib.ha2g_se_bwd_scale_b16(dy16.data_ptr(), x16.data_ptr(), y16.data_ptr(), ds.data_ptr(), B, H * W, C, s.data_ptr(), ws.data_ptr(), _stream()))),
    ]
    for name, nbytes, f32, f16 in tests:
        a, b = t_us(f32), t_us(f16)
        print('C=%-3d %3dx%-3d  %-30s %9.1f %9.0f %9.1f %9.0f' % (C, H, W, name, a, nbytes / a / 1e3, b, nbytes / 2 / b / 1e3))

"""Accuracy and speed of the split-bf16 (bf16x3) core vs exact fp32 MFMA on GEMM / conv shapes of the train step."""
import sys, torch
sys.path.insert(0, '.')
from ha2g_amd import ops, wav_engine as we
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
torch.manual_seed(0)
for (M, N, K) in ((4352, 900, 600), (13056, 900, 600), (4096, 4096, 4096), (13056, 300, 600)):
    a, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    ref = (a.double() @ b.double().t())
    out = {}
    for mode in (0, 1):
        lib.ha2g_gemm_set_mode(mode)
        c = ops.gemm(a, b, transb=True)
        err = float((c.double() - ref).abs().max() / ref.abs().max())
        rms = float(((c.double() - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())
        us = timeit(lambda: ops.gemm(a, b, transb=True))
        out[mode] = (err, rms, us)
    print('gemm %dx%dx%d: fp32 max-rel %.1e rms-rel %.1e %.0fus %.0fTF | x3 max-rel %.1e rms-rel %.1e %.0fus %.0fTF' % (
        M, N, K, out[0][0], out[0][1], out[0][2], 2.0 * M * N * K / out[0][2] / 1e6, out[1][0], out[1][1], out[1][2], 2.0 * M * N * K / out[1][2] / 1e6))
B = 128
for H, W, C in ((128, 70, 32), (64, 35, 64), (32, 18, 128), (16, 9, 256)):
    x = torch.randn(B, H, W, C, device=dev).clamp_min(0) * 2; w = torch.randn(C, 3, 3, C, device=dev) * (2.0 / (9 * C)) ** 0.5
    fl = 2.0 * B * H * W * C * C * 9
    res = {}
    for mode in (0, 1):
        lib.ha2g_gemm_set_mode(mode)
        y = we.conv_fwd(x, w, None, 1, 1, 0)
        us = timeit(lambda: we.conv_fwd(x, w, None, 1, 1, 0))
        res[mode] = (y, us)
    d = (res[1][0].double() - res[0][0].double())
    print('conv C=%d %dx%d: fp32 %.0fus %.0fTF | x3 %.0fus %.0fTF | x3 vs fp32 max-rel %.1e rms-rel %.1e' % (
        C, H, W, res[0][1], fl / res[0][1] / 1e6, res[1][1], fl / res[1][1] / 1e6, float(d.abs().max() / res[0][0].abs().max()),
        float((d ** 2).mean().sqrt() / (res[0][0].double() ** 2).mean().sqrt())))
lib.ha2g_gemm_set_mode(6)

"""Accuracy and speed of the split-bf16 cores (x3 = 2 pieces / 3 MFMAs, x6 = 3 pieces / 6 MFMAs) vs the fp32 MFMA on forward
GEMM / conv shapes of the train step."""
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
for (M, N, K) in ((4352, 900, 600), (13056, 900, 600), (13056, 900, 348), (4096, 4096, 4096), (13056, 300, 600), (13056, 32, 600), (4352, 64, 192)):
    a, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    ref = (a.double() @ b.double().t())
    out = {}
    for mode in (0, 1, 8):
        lib.ha2g_gemm_set_mode(mode)
        c = ops.gemm(a, b, transb=True)
        err = float((c.double() - ref).abs().max() / ref.abs().max())
        rms = float(((c.double() - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())
        us = timeit(lambda: ops.gemm(a, b, transb=True))
        out[mode] = (err, rms, us)
    print('gemm %dx%dx%d: ' % (M, N, K) + ' | '.join('%s max-rel %.1e rms-rel %.1e %.0fus %.0fTF' % (
        nm, out[m][0], out[m][1], out[m][2], 2.0 * M * N * K / out[m][2] / 1e6) for m, nm in ((0, 'fp32'), (1, 'x3'), (8, 'x6'))))
B = 128
for H, W, C in ((128, 70, 32), (64, 35, 64), (32, 18, 128), (16, 9, 256)):
    x = torch.randn(B, H, W, C, device=dev).clamp_min(0) * 2; w = torch.randn(C, 3, 3, C, device=dev) * (2.0 / (9 * C)) ** 0.5
    fl = 2.0 * B * H * W * C * C * 9
    res = {}
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2)[:8].double(), w.permute(0, 3, 1, 2).double(), padding=1).permute(0, 2, 3, 1)
    for mode in (0, 1, 8):
        lib.ha2g_gemm_set_mode(mode)
        y = we.conv_fwd(x, w, None, 1, 1, 0)
        us = timeit(lambda: we.conv_fwd(x, w, None, 1, 1, 0))
        res[mode] = (y, us)
    def rms(y):
        d = y[:8].double() - ref
        return float((d ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())
    print('conv C=%d %dx%d: ' % (C, H, W) + ' | '.join('%s %.0fus %.0fTF rms-rel vs f64 %.1e' % (nm, res[m][1], fl / res[m][1] / 1e6, rms(res[m][0]))
                                                        for m, nm in ((0, 'fp32'), (1, 'x3'), (8, 'x6'))))
lib.ha2g_gemm_set_mode(6)

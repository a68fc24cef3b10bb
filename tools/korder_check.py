"""Plane kernel, channel-major vs tap-major k order, against float64 (forward and data gradient, the tower's shapes at B = 16)."""
import sys
import torch
sys.path.insert(0, '.')
from ha2g_amd import ops, wav_engine as we
from ha2g_amd._lib import lib
import torch.nn.functional as F
dev = torch.device('cuda:0')
torch.manual_seed(0)
B = 16
for (Cin, Cout, K, stride, pad, H, W) in ((64, 64, 3, 1, 1, 32, 18), (128, 128, 3, 1, 1, 16, 9), (256, 256, 3, 1, 1, 8, 5), (32, 64, 3, 2, 1, 64, 35), (32, 64, 1, 2, 0, 64, 35),
                                          (64, 128, 3, 2, 1, 32, 18), (128, 256, 3, 2, 1, 16, 9), (64, 64, 2, 1, 0, 32, 18), (64, 64, 3, 1, 1, 64, 35)):
    x = torch.randn(B, H, W, Cin, device=dev)
    w = torch.randn(Cout, K, K, Cin, device=dev) * 0.05
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), stride=stride, padding=pad).permute(0, 2, 3, 1)
    OH, OW = ref.shape[1:3]
    dy = torch.randn(B, OH, OW, Cout, device=dev)
    refd = torch.autograd.grad(F.conv2d(x.double().permute(0, 3, 1, 2).requires_grad_(True), w.double().permute(0, 3, 1, 2), stride=stride, padding=pad),
                               [], []) if False else None
    xd = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    yd = F.conv2d(xd, w.double().permute(0, 3, 1, 2), stride=stride, padding=pad)
    refdx = torch.autograd.grad(yd, xd, dy.double().permute(0, 3, 1, 2))[0].permute(0, 2, 3, 1)
    row = []
    for kmaj in (0, 1):
        lib.ha2g_conv_planes_korder(kmaj)
        e_f = None
        if we.fwd_planes_ok(w, stride, pad):
            xp = ops.to_planes(x)
            wpl = torch.empty(3, Cout, K, K, Cin, dtype=torch.bfloat16, device=dev)
            ops.check(lib.ha2g_f32_to_planes_np(w.data_ptr(), wpl.data_ptr(), wpl.stride(0), 3, w.numel(), torch.cuda.current_stream().cuda_stream))
            y = we.conv_fwd_planes(xp, wpl, (B, H, W, Cin), stride, pad, ops.ACT_NONE)
            e_f = float((y.double() - ref).abs().max() / ref.abs().max())
        try:
            dx = we.conv_dgrad_planes(ops.to_planes(dy), w, (B, H, W, Cin), stride, pad)
            e_d = float((dx.double() - refdx).abs().max() / refdx.abs().max())
        except Exception:
            e_d = float('nan')                             # geometry the plane data-gradient kernel does not serve
        row.append((e_f, e_d))
    print('Cin %3d Cout %3d k %d s %d p %d %2dx%-2d   fwd err tap-major %s channel-major %s   dgrad err tap-major %.2e channel-major %.2e' % (
        Cin, Cout, K, stride, pad, H, W, '%.2e' % row[0][0] if row[0][0] is not None else '   -    ', '%.2e' % row[1][0] if row[1][0] is not None else '   -    ', row[0][1], row[1][1]))
lib.ha2g_conv_planes_korder(1)

"""The plane kernel (pconv_q_kernel, three pieces) alone at the trunk's three shapes, for PMC passes (tools/pmc_q.sh).  usage: python tools/q_kernel_run.py [iters]"""
import sys
import torch
sys.path.insert(0, '.')
from ha2g_amd import ops, wav_engine as we
from ha2g_amd._lib import lib

dev = torch.device('cuda:0')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 128
for H, W, C in ((64, 35, 64), (32, 18, 128), (16, 9, 256)):
    dy = torch.randn(B, H, W, C, device=dev)
    w = torch.randn(C, 3, 3, C, device=dev) * 0.05
    out = torch.empty(B, H, W, C, device=dev)
    pl = ops.to_planes(dy)
    wpl = we.weight_planes(w, pl.shape[0])
    for _ in range(iters):
        lib.ha2g_conv2d_dgrad_planes_np_f32(pl.data_ptr(), pl.stride(0), wpl.data_ptr(), wpl.stride(0), pl.shape[0], out.data_ptr(), B, H, W, C, C, 3, 3, 1, 1, 0.0,
                                            torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()

"""The GRU input projection GEMM alone (13056 x 900 x 600, fp32 MFMA), for PMC passes (tools/pmc_gemm_gi.sh)."""
import sys, torch
sys.path.insert(0, '.')
from ha2g_amd import ops
a = torch.randn(13056, 600, device='cuda:0'); b = torch.randn(900, 600, device='cuda:0')
for _ in range(12): ops.gemm(a, b, transb=True)
torch.cuda.synchronize()

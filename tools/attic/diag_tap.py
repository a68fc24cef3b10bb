import sys
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from ha2g_amd import ops, wav_engine as we
from ha2g_amd._lib import lib, check
DEV = 'cuda:0'
torch.manual_seed(0)
N, H, W, C, k = 3, 64, 35, 64, 2
r = np.random.Generator(np.random.PCG64(3))
f = torch.from_numpy(r.standard_normal((N, C, H, W)).astype(np.float32)).clamp_min(0) * 3
w = torch.from_numpy((r.standard_normal((C, C, k, k)) / (C * k * k) ** 0.5).astype(np.float32))
b = torch.from_numpy((0.1 * r.standard_normal(C)).astype(np.float32))
g, be = torch.from_numpy((1 + 0.1 * r.standard_normal(C)).astype(np.float32)), torch.from_numpy((0.1 * r.standard_normal(C)).astype(np.float32))
OH, OW = H - k + 1, W - k + 1
dat = torch.from_numpy(r.standard_normal((N, C, OH, OW)).astype(np.float32))
# CPU fp64
f64, w64, b64 = f.double(), w.double().requires_grad_(True), b.double().requires_grad_(True)
c = F.conv2d(f64, w64, b64)
c.retain_grad()
a = F.batch_norm(torch.relu(c), None, None, g.double(), be.double(), True, 0.1, 1e-5)
(a * dat.double()).sum().backward()
# GPU pieces
fg = f.permute(0, 2, 3, 1).contiguous().to(DEV)
wg = w.permute(0, 2, 3, 1).contiguous().to(DEV)
ct = we.conv_fwd(fg, wg, b.to(DEV), 1, 0, 1)
rows = ct.view(-1, C)
mean, invstd = ops.bn_stats(rows, None, None, 0.1, 1e-5)
at = ops.bn_apply(rows, mean, invstd, g.to(DEV), be.to(DEV))
datg = dat.permute(0, 2, 3, 1).contiguous().to(DEV).view(-1, C)
dct, dg, db = ops.bn_bwd(datg, rows, mean, invstd, g.to(DEV))
dct = ops.eltwise(ops.OP_RELU_BWD, dct, rows).view(ct.shape)
def rel(x, ref):
    ref = ref.double(); return float((x.double().cpu() - ref).abs().max() / ref.abs().max())
print('ct', rel(ct.permute(0, 3, 1, 2), torch.relu(c).detach()))
print('at', rel(at.view(ct.shape).permute(0, 3, 1, 2), a.detach()))
print('dct', rel(dct.permute(0, 3, 1, 2), c.grad))
print('dbias', rel(ops.colsum(dct.view(-1, C)), b64.grad))
print('dw', rel(we.conv_wgrad(fg, dct, wg, 1, 0), w64.grad))

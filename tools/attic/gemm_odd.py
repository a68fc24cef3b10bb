"""GEMM accuracy sweep over odd / unaligned shapes (K, N, leading dimensions not multiples of 2 or 4) vs float64."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from ha2g_amd import ops
from ha2g_amd._lib import lib

dev = 'cuda:0'
r = np.random.Generator(np.random.PCG64(1))


def rnd(*s):
    return torch.from_numpy(r.standard_normal(s).astype(np.float32))


def run(M, N, K, ta, tb, beta, mode, bias=False):
    a = rnd(K, M) if ta else rnd(M, K)
    b = rnd(N, K) if tb else rnd(K, N)
    c0 = rnd(M, N)
    bi = rnd(N) if bias else None
    ref = (a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double()) + beta * c0.double()
    if bias:
        ref = ref + bi.double()
    lib.ha2g_gemm_set_mode(mode)
    out = c0.clone().to(dev)
    ops.gemm(a.to(dev), b.to(dev), transa=ta, transb=tb, out=out, beta=beta, bias=bi.to(dev) if bias else None)
    lib.ha2g_gemm_set_mode(6)
    err = (out.double().cpu() - ref).abs().max() / ref.abs().max()
    return float(err)


for mode in (0, 6):
    for (M, N, K, ta, tb, beta) in [
        (136, 900, 105, False, True, 0.0), (136, 900, 111, False, True, 0.0), (136, 900, 108, False, True, 0.0), (136, 900, 102, False, True, 0.0),
        (136, 105, 900, False, False, 0.0), (136, 111, 900, False, False, 1.0), (136, 108, 900, False, False, 1.0), (136, 207, 900, False, False, 1.0),
        (900, 105, 136, True, False, 0.0), (900, 111, 136, True, False, 1.0), (900, 108, 136, True, False, 1.0), (900, 207, 4352, True, False, 1.0),
        (4352, 207, 900, False, False, 1.0), (4352, 900, 207, False, True, 0.0), (150, 126, 4352, True, False, 0.0), (4352, 126, 150, False, True, 0.0),
        (4352, 150, 126, False, False, 0.0), (4352, 16, 378, False, True, 0.0), (4352, 378, 16, False, False, 0.0), (16, 378, 4352, True, False, 0.0),
    ]:
        e = run(M, N, K, ta, tb, beta, mode, bias=not ta and tb)
        flag = '  <<<<' if e > 2e-5 else ''
        print('mode %d M=%5d N=%4d K=%5d ta=%d tb=%d beta=%.0f  err %.2e%s' % (mode, M, N, K, ta, tb, beta, e, flag))

"""CPU checks of the oracle's bf16-storage rounding points (oracle.ha2g_oracle.bf16_storage, the checker of tests/test_gpu_b16.py):
switched off they are identities (the pinned fp32 / fp64 oracle is untouched, bit for bit), switched on every tensor the mode stores is
bf16-representable, parameter gradients stay fp32-valued, and the format's cost on one well-conditioned SEBasicBlock is what bf16 storage
should cost (a few 1e-3 on the output).  Reference being restated: scripts/model/ResNetBlocks.py:21-37,81-95."""
import torch

from ha2g_amd.config import BLOCK_B, BLOCK_CASES, BLOCK_SEED
from ha2g_testing import block_io, block_state
from oracle import ha2g_oracle as O


def _run(name, b16, dt=torch.float32):
    geom = BLOCK_CASES[name]
    sd = block_state(name, geom, BLOCK_SEED, dt)
    x, wl = block_io(name, geom, BLOCK_B, BLOCK_SEED, dt)
    x = x.to(torch.bfloat16).to(dt).requires_grad_(True)
    ps = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and not k.endswith(('running_mean', 'running_var'))}
    if b16:
        with O.bf16_storage():
            y = O.se_block(x, sd, '', 2 if geom[4] else 1, geom[4])
            grads = torch.autograd.grad((y * wl).sum(), [x] + list(ps.values()))
    else:
        y = O.se_block(x, sd, '', 2 if geom[4] else 1, geom[4])
        grads = torch.autograd.grad((y * wl).sum(), [x] + list(ps.values()))
    return y.detach(), grads, sd


def _is_bf16(t):
    return torch.equal(t, t.to(torch.bfloat16).to(t.dtype))


def test_rounding_points_are_identities_when_switched_off():
    assert not O._BF16_STORAGE[0]
    t = torch.randn(1000)
    assert O._st(t) is t and O._wq(t) is t and O._gr(t) is t
    with O.bf16_storage():
        assert O._BF16_STORAGE[0]
        assert _is_bf16(O._st(t)) and _is_bf16(O._wq(t)) and torch.equal(O._gr(t), t)
    assert not O._BF16_STORAGE[0]
    y0, g0, _ = _run('l3d', False)
    y1, g1, _ = _run('l3d', False)
    assert torch.equal(y0, y1) and all(torch.equal(a, b) for a, b in zip(g0, g1))


def test_stored_tensors_are_bf16_and_parameter_gradients_are_not():
    for name in ('l2', 'l3d'):
        y32, g32, _ = _run(name, False)
        y16, g16, sd = _run(name, True)
        assert _is_bf16(y16) and not _is_bf16(y32)                       # the block output is a stored tensor
        assert not _is_bf16(g16[1])                                      # conv1.weight gradient: fp32 accumulation, not rounded
        rel = float((y16 - y32).norm() / y32.norm())
        assert 1e-4 < rel < 1e-2, rel                                    # what bf16 storage costs on one block's output
        for a, b in zip(g16[1:], g32[1:]):
            assert torch.isfinite(a).all()
            assert float((a - b).norm() / b.norm()) < 0.3
        assert all(torch.isfinite(v).all() for k, v in sd.items() if k.endswith(('running_mean', 'running_var')))


def test_gradient_stream_is_rounded_once_per_stored_tensor():
    """_StoreBf16 / _GradBf16 round the incoming gradient, _WeightBf16 does not: the three behaviours the rounding points are built from."""
    g = torch.randn(64) * 3.0
    for fn, rounds in ((O._StoreBf16, True), (O._GradBf16, True), (O._WeightBf16, False)):
        x = torch.randn(64, requires_grad=True)
        (gx,) = torch.autograd.grad((fn.apply(x) * g).sum(), x)
        assert torch.equal(gx, g.to(torch.bfloat16).float() if rounds else g)

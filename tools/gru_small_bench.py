"""Discriminator-sized GRU layer (H=64, T=28, B=128) forward / BPTT launch times."""
import sys, torch
sys.path.insert(0, '.')
from ha2g_amd._lib import lib, check
dev = torch.device('cuda:0')
def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for B, T, H in ((128, 28, 64), (256, 28, 64), (4, 28, 32)):
    st = torch.cuda.current_stream().cuda_stream
    gi = torch.randn(B * T, 6 * H, device=dev); whh = torch.randn(3 * H, H, device=dev) / H ** 0.5
    npk = lib.ha2g_gru_packed_floats(H); pk = torch.empty(4, npk, device=dev)
    check(lib.ha2g_gru_pack_whh(whh.data_ptr(), pk[0].data_ptr(), pk[2].data_ptr(), H, st)); check(lib.ha2g_gru_pack_whh(whh.data_ptr(), pk[1].data_ptr(), pk[3].data_ptr(), H, st))
    bhh = torch.randn(3 * H, device=dev); y = torch.empty(B, T, 2 * H, device=dev); rs = torch.empty(B, T, 2, 4, H, device=dev)
    dg = torch.empty(B * T, 8 * H, device=dev); dy = torch.randn(B, T, 2 * H, device=dev); hp = torch.empty(B, T, 2 * H, device=dev)
    f = timeit(lambda: check(lib.ha2g_gru_layer_fwd(gi.data_ptr(), pk.data_ptr(), bhh.data_ptr(), bhh.data_ptr(), y.data_ptr(), rs.data_ptr(), B, T, H, st)))
    b = timeit(lambda: check(lib.ha2g_gru_layer_bwd(dy.data_ptr(), y.data_ptr(), rs.data_ptr(), pk[2].data_ptr(), dg.data_ptr(), hp.data_ptr(), B, T, H, st)))
    print('B=%d T=%d H=%d: fwd %.1f us (%.2f us/step)  bwd %.1f us (%.2f us/step)' % (B, T, H, f, f / T, b, b / T))

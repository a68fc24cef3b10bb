import sys, ctypes, torch
sys.path.insert(0, '.')
from ha2g_amd import ops
from ha2g_amd._lib import lib, check, LIB_PATH
raw = ctypes.CDLL(LIB_PATH)
dev = torch.device('cuda:0')
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
B, T, H = 384, 34, 300
st = torch.cuda.current_stream().cuda_stream
gi = torch.randn(B * T, 6 * H, device=dev); whh = torch.randn(3 * H, H, device=dev) / H ** 0.5
npk = lib.ha2g_gru_packed_floats(H); pk = torch.empty(4, npk, device=dev)
check(lib.ha2g_gru_pack_whh(whh.data_ptr(), pk[0].data_ptr(), pk[2].data_ptr(), H, st)); check(lib.ha2g_gru_pack_whh(whh.data_ptr(), pk[1].data_ptr(), pk[3].data_ptr(), H, st))
bhh = torch.randn(3 * H, device=dev); y = torch.empty(B, T, 2 * H, device=dev); rs = torch.empty(B, T, 2, 4, H, device=dev)
xch, err = ops._cluster_scratch(dev)
for mode, name in ((0, 'normal (L2-scope publish when the cluster shares an XCD)'), (4, 'write-through publish forced'), (1, 'publish+one gather pass, no wait'), (2, 'no publish, no gather')):
    raw.ha2g_gru_cluster_debug(mode)
    us = timeit(lambda: check(lib.ha2g_gru_layer_fwd_cluster(gi.data_ptr(), pk.data_ptr(), bhh.data_ptr(), bhh.data_ptr(), y.data_ptr(), rs.data_ptr(), xch.data_ptr(), err.data_ptr(), B, T, H, st)))
    print('%-60s %.1f us  %.2f us/step' % (name, us, us / T))
raw.ha2g_gru_cluster_debug(0)
dg = torch.empty(B * T, 8 * H, device=dev); dy = torch.randn(B, T, 2 * H, device=dev)
for Bb in (128, 384):
    for mode, name in ((0, 'bwd normal'), (4, 'bwd write-through forced'), (1, 'bwd no wait'), (2, 'bwd no exchange')):
        raw.ha2g_gru_cluster_debug(mode)
        us = timeit(lambda: check(lib.ha2g_gru_layer_bwd_cluster(dy.data_ptr(), y.data_ptr(), rs.data_ptr(), pk[2].data_ptr(), dg.data_ptr(), 0, xch.data_ptr(), err.data_ptr(), Bb, T, H, st)))
        print('B=%d %-55s %.1f us  %.2f us/step' % (Bb, name, us, us / T))
raw.ha2g_gru_cluster_debug(0)
print('err', ops.gru_cluster_error(dev))

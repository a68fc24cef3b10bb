"""Forward cluster GRU layer alone (B=384, T=34), 30 launches: the target of PMC counter passes (tools/pmc_gru_kernel.sh)."""
import sys, torch
sys.path.insert(0, '.')
from ha2g_amd import ops
from ha2g_amd._lib import lib, check
dev = torch.device('cuda:0')
B, T, H = 384, 34, 300
st = torch.cuda.current_stream().cuda_stream
gi = torch.randn(B * T, 6 * H, device=dev); whh = torch.randn(3 * H, H, device=dev) / H ** 0.5
npk = lib.ha2g_gru_packed_floats(H); pk = torch.empty(4, npk, device=dev)
check(lib.ha2g_gru_pack_whh(whh.data_ptr(), pk[0].data_ptr(), pk[2].data_ptr(), H, st)); check(lib.ha2g_gru_pack_whh(whh.data_ptr(), pk[1].data_ptr(), pk[3].data_ptr(), H, st))
bhh = torch.randn(3 * H, device=dev); y = torch.empty(B, T, 2 * H, device=dev); rs = torch.empty(B, T, 2, 4, H, device=dev)
xch, err = ops._cluster_scratch(dev)
dg = torch.empty(B * T, 8 * H, device=dev); dy = torch.randn(B, T, 2 * H, device=dev); hp = torch.empty(B, T, 2 * H, device=dev)
for _ in range(30):
    check(lib.ha2g_gru_layer_fwd_cluster(gi.data_ptr(), pk.data_ptr(), bhh.data_ptr(), bhh.data_ptr(), y.data_ptr(), rs.data_ptr(), xch.data_ptr(), err.data_ptr(), B, T, H, st))
    check(lib.ha2g_gru_layer_bwd_cluster(dy.data_ptr(), y.data_ptr(), rs.data_ptr(), pk[2].data_ptr(), dg.data_ptr(), hp.data_ptr(), xch.data_ptr(), err.data_ptr(), 128, T, H, st))
torch.cuda.synchronize()
print('err', ops.gru_cluster_error(dev))

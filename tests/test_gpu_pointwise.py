"""Direct float64 checks of the HIP kernels that the whole-module fixtures only reach indirectly: the SE tail
(forward, squeeze gradient, apply gradient), PixelShuffle (both directions), the tap flatten, the speaker-softmax blend,
the embedding gradient, weight_norm, Adam, and the Philox dropout (keep rate, scaling, mask placement)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def rnd(shape, seed, scale=1.0):
    r = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((scale * r.standard_normal(shape)).astype(np.float32))


def relerr(got, ref):
    ref = ref.double()
    return float((got.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('shape', [(3, 35, 32), (2, 576, 64), (5, 40, 128), (1, 144, 256), (4, 8960, 32)])
def test_se_tail_fwd_bwd(shape):
    """out = relu(x * s[n,c] + res) (ResNetBlocks.py:33-36,91-95); ds = sum_hw dout*(out>0)*x; dres, dx (with the
    squeeze's gradient dpool folded in)."""
    from ha2g_amd._lib import check, lib
    from ha2g_amd.ops import _stream
    N, HW, C = shape
    x, res, dout = rnd((N, HW, C), 1), rnd((N, HW, C), 2), rnd((N, HW, C), 3)
    s = torch.sigmoid(rnd((N, C), 4))
    dpool = rnd((N, C), 5, 0.1)
    xd, rd, sd_, dd = x.double().requires_grad_(True), res.double().requires_grad_(True), s.double().requires_grad_(True), dout.double()
    out64 = torch.relu(xd * sd_[:, None, :] + rd)
    gx, gr, gs = torch.autograd.grad((out64 * dd).sum(), [xd, rd, sd_])
    gx = gx + dpool.double()[:, None, :]
    xg, rg, sg, dg, pg = (t.to(DEV).contiguous() for t in (x, res, s, dout, dpool))
    out, ds = torch.empty_like(xg), torch.empty(N, C, device=DEV)
    dres, dx = torch.empty_like(xg), torch.empty_like(xg)
    check(lib.ha2g_se_scale_add_relu_f32(xg.data_ptr(), sg.data_ptr(), rg.data_ptr(), out.data_ptr(), N, HW, C, _stream()))
    from ha2g_amd import ops as _ops
    check(lib.ha2g_se_bwd_scale_f32(dg.data_ptr(), out.data_ptr(), xg.data_ptr(), ds.data_ptr(), N, HW, C, None, _ops.workspace(xg.device).data_ptr(), _stream()))
    ds1 = torch.empty_like(ds)
    check(lib.ha2g_se_bwd_scale_f32(dg.data_ptr(), out.data_ptr(), xg.data_ptr(), ds1.data_ptr(), N, HW, C, None, None, _stream()))       # one block per image
    assert relerr(ds1, gs) < 2e-6
    for ws_ in (_ops.workspace(xg.device).data_ptr(), None):                                     # gate folded in: ds * s (1 - s)
        dsg = torch.empty_like(ds)
        check(lib.ha2g_se_bwd_scale_f32(dg.data_ptr(), out.data_ptr(), xg.data_ptr(), dsg.data_ptr(), N, HW, C, sg.data_ptr(), ws_, _stream()))
        assert relerr(dsg, gs * sd_.detach() * (1 - sd_.detach())) < 2e-6
    check(lib.ha2g_se_bwd_apply_f32(dg.data_ptr(), out.data_ptr(), sg.data_ptr(), pg.data_ptr(), dres.data_ptr(), dx.data_ptr(), N, HW, C, _stream()))
    assert relerr(out, out64.detach()) < 2e-7
    assert relerr(ds, gs) < 2e-6
    assert relerr(dres, gr) < 2e-7
    assert relerr(dx, gx) < 2e-7


@pytest.mark.parametrize('case', [(2, 32, 18, 128, 2), (3, 16, 9, 256, 4), (1, 5, 3, 64, 2)])
def test_pixel_shuffle_both_directions(case):
    from ha2g_amd import wav_engine as we
    N, H, W, C, r = case
    x = rnd((N, C, H, W), 7)
    ref = F.pixel_shuffle(x, r)
    got = we._pixel_shuffle(x.permute(0, 2, 3, 1).contiguous().to(DEV), r)
    assert torch.equal(got.permute(0, 3, 1, 2).cpu(), ref)
    dy = rnd(tuple(ref.shape), 8)
    back = we._pixel_shuffle(dy.permute(0, 2, 3, 1).contiguous().to(DEV), r, inverse=True, shape=(N, H, W, C))
    assert torch.equal(back.permute(0, 3, 1, 2).cpu(), F.pixel_unshuffle(dy, r))


@pytest.mark.parametrize('case', [(2, 63, 34, 64), (3, 62, 6, 32), (1, 62, 34, 16), (2, 5, 3, 8)])
def test_tap_flatten_both_directions(case):
    """[N,H,W,C] -> rows (n,w), K index c*H+h == reshape(B, C*H, W).transpose(1,2) of the reference (ResNetSE34V2.py:160-162)."""
    from ha2g_amd import wav_engine as we
    N, H, W, C = case
    x = rnd((N, C, H, W), 9)
    ref = x.reshape(N, C * H, W).transpose(1, 2).reshape(N * W, C * H)
    got = we._tap_pack(x.permute(0, 2, 3, 1).contiguous().to(DEV))
    assert torch.equal(got.cpu(), ref)
    d = rnd((N * W, C * H), 10)
    back = we._tap_pack(d.to(DEV), inverse=True, shape=(N, H, W, C))
    assert torch.equal(back.permute(0, 3, 1, 2).cpu(), d.view(N, W, C * H).transpose(1, 2).reshape(N, C, H, W))


@pytest.mark.parametrize('L', [3, 6])
def test_blend_fwd_bwd(L):
    from ha2g_amd._lib import check, lib
    from ha2g_amd.ops import _stream
    B, T = 5, 34
    logits = rnd((B, 3 * L), 11)
    f = [rnd((B, T, 32), 12 + i) for i in range(3)]
    dbl, dw_ext = rnd((L, B, T, 32), 20), rnd((B, 3, L), 21)
    df0 = [rnd((B, T, 32), 30 + i) for i in range(3)]
    lg = logits.double().requires_grad_(True)
    fd = [t.double().requires_grad_(True) for t in f]
    w64 = lg.view(B, 3, L).softmax(1)
    bl64 = torch.stack([fd[0] * w64[:, 0, i, None, None] + fd[1] * w64[:, 1, i, None, None] + fd[2] * w64[:, 2, i, None, None] for i in range(L)])
    g = torch.autograd.grad((bl64 * dbl.double()).sum() + (w64 * dw_ext.double()).sum(), [lg] + fd)
    lgg = logits.to(DEV)
    fg = [t.to(DEV) for t in f]
    w, bl = torch.empty(B, 3, L, device=DEV), torch.empty(L, B, T, 32, device=DEV)
    check(lib.ha2g_blend_fwd_f32(lgg.data_ptr(), fg[0].data_ptr(), fg[1].data_ptr(), fg[2].data_ptr(), w.data_ptr(), bl.data_ptr(), B, L, T * 32, _stream()))
    assert relerr(w, w64.detach()) < 5e-7 and relerr(bl, bl64.detach()) < 5e-7
    df = [t.to(DEV).clone() for t in df0]
    dlog = torch.empty(B, 3 * L, device=DEV)
    check(lib.ha2g_blend_bwd_f32(dbl.to(DEV).data_ptr(), dw_ext.to(DEV).data_ptr(), w.data_ptr(), fg[0].data_ptr(), fg[1].data_ptr(), fg[2].data_ptr(),
                                 df[0].data_ptr(), df[1].data_ptr(), df[2].data_ptr(), dlog.data_ptr(), B, L, T * 32, _stream()))
    assert relerr(dlog, g[0]) < 5e-6
    for i in range(3):                                       # the kernel ADDS the blend's share onto the incoming tap gradients
        assert relerr(df[i], g[1 + i] + df0[i].double()) < 1e-6


def test_embedding_bwd_duplicates_and_padding():
    from ha2g_amd import ops
    V, C, B, T = 50, 300, 9, 34
    r = np.random.Generator(np.random.PCG64(3))
    tok = np.zeros((B, T), np.int64)
    for b in range(B):
        pos = r.choice(T, size=7, replace=False)
        tok[b, pos] = r.integers(1, 6, size=7)                # few distinct ids => many duplicates
    tok[0, :] = 7                                             # one id 34 times in a row
    w = rnd((V, C), 4).to(DEV).requires_grad_(True)
    dy = rnd((B, T, C), 5)
    out = ops.embedding(torch.from_numpy(tok).to(DEV), w)
    assert torch.equal(out.detach().cpu(), w.detach().cpu()[torch.from_numpy(tok)])
    out.backward(dy.to(DEV))
    ref = torch.zeros(V, C, dtype=torch.float64).index_add_(0, torch.from_numpy(tok).reshape(-1), dy.double().reshape(-1, C))
    assert relerr(w.grad, ref) < 1e-6
    assert float(w.grad[8:].abs().max()) == 0.0


@pytest.mark.parametrize('shape', [(300, 300, 2), (16, 27, 3), (5, 3, 1)])
def test_weight_norm_fwd_bwd(shape):
    from ha2g_amd import ops
    v = rnd(shape, 6, 0.01)
    g = (0.5 + torch.rand(shape[0], 1, 1, generator=torch.Generator().manual_seed(1)))
    dw = rnd(shape, 7)
    vd, gd = v.double().requires_grad_(True), g.double().requires_grad_(True)
    w64 = gd * vd / vd.flatten(1).norm(dim=1).view(-1, 1, 1)
    gg, gv = torch.autograd.grad((w64 * dw.double()).sum(), [gd, vd])
    vg, ggpu = v.to(DEV).requires_grad_(True), g.to(DEV).requires_grad_(True)
    w = ops.weight_norm(ggpu, vg)
    w.backward(dw.to(DEV))
    assert relerr(w, w64.detach()) < 5e-7
    assert relerr(ggpu.grad, gg) < 5e-6
    assert relerr(vg.grad, gv) < 5e-6


def test_adam_matches_torch_adam_fp64():
    """Five steps of ha2g_adam_f32 (through FusedAdam) against torch.optim.Adam in float64 (train.py:155-170 settings),
    odd element counts (scalar tail) and a zero-gradient tensor (moments decay, parameter unchanged)."""
    from ha2g_amd.optim import FusedAdam
    shapes = [(37, 5), (1,), (300, 300, 2), (7,)]
    ps = [torch.nn.Parameter(rnd(s, 40 + i).to(DEV)) for i, s in enumerate(shapes)]
    ref = [torch.nn.Parameter(p.detach().double().cpu()) for p in ps]
    opt = FusedAdam(ps, lr=5e-4, betas=(0.5, 0.999))
    ropt = torch.optim.Adam(ref, lr=5e-4, betas=(0.5, 0.999))
    for step in range(5):
        opt.zero_grad()
        for i, (p, q) in enumerate(zip(ps, ref)):
            g = rnd(tuple(p.shape), 100 * step + i) if i != 3 else torch.zeros(p.shape)
            p.grad.copy_(g.to(DEV))
            q.grad = g.double()
        opt.step()
        ropt.step()
        for p, q in zip(ps, ref):
            # O(1) parameters move by lr = 5e-4 per step: the fp32 parameter's own rounding (half an ulp per update) dominates
            assert float((p.detach().double().cpu() - q.detach()).abs().max()) < (1.2e-7 * float(q.abs().max()) + 1e-3 * 5e-4) * (step + 1)
        off = 0
        for q in ref:                                             # the moments are the sharper check of the update arithmetic
            st = ropt.state[q]
            n = q.numel()
            assert relerr(opt.flat_m[off:off + n], st['exp_avg'].reshape(-1)) < 2e-6
            assert relerr(opt.flat_v[off:off + n], st['exp_avg_sq'].reshape(-1)) < 2e-6
            off += (n + 3) // 4 * 4
    assert torch.equal(ps[3].detach().cpu(), rnd((7,), 43))


def test_dropout_kernel_statistics_and_masks():
    """Philox dropout: keep rate within 4 sigma of 1-p, kept values scaled by exactly 1/(1-p), backward uses the forward's
    mask, masks differ between call sites, steps and seeds (= ranks), and are reproducible for equal (seed, step, site)."""
    from ha2g_amd import ops
    n = 1 << 20
    x = (rnd((n,), 1).abs() + 0.5).to(DEV)
    for p in (0.1, 0.3):
        ops.rng.seed(torch.device(DEV), 77)
        ops.rng.begin_step()
        xr = x.clone().requires_grad_(True)
        y = ops.dropout(xr, p, True)
        keep = y != 0
        rate = float(keep.float().mean())
        assert abs(rate - (1 - p)) < 4 * np.sqrt(p * (1 - p) / n), (p, rate)
        assert torch.equal(y[keep], (x * np.float32(1.0 / (1.0 - np.float32(p))))[keep])            # exact 1/(1-p) scaling
        y.backward(torch.ones_like(y))
        assert torch.equal(xr.grad != 0, keep)                                                  # same mask in backward
        assert torch.allclose(xr.grad[keep], torch.full_like(xr.grad[keep], 1.0 / (1.0 - p)))
        y2 = ops.dropout(x, p, True)                                                            # next call site, same step
        assert float(((y2 != 0) == keep).float().mean()) < 1 - 2 * p * (1 - p) + 0.01
        ops.rng.end_step()
        ops.rng.begin_step()
        y3 = ops.dropout(x, p, True)                                                            # same site, next step
        assert float(((y3 != 0) == keep).float().mean()) < 1 - 2 * p * (1 - p) + 0.01
        ops.rng.seed(torch.device(DEV), 78)                                                     # another rank's seed
        y4 = ops.dropout(x, p, True)
        assert float(((y4 != 0) == keep).float().mean()) < 1 - 2 * p * (1 - p) + 0.01
        ops.rng.seed(torch.device(DEV), 77)                                                     # replay: identical
        assert torch.equal(ops.dropout(x, p, True), y)
    assert ops.dropout(x, 0.3, False) is x and ops.dropout(x, 0.0, True) is x
    # the mask is re-drawn in the backward (not stored): a backward issued after the step counter advanced must fail loudly
    ops.rng.begin_step()
    xr = x.clone().requires_grad_(True)
    y5 = ops.dropout(xr, 0.3, True)
    ops.rng.end_step()
    with pytest.raises(RuntimeError):
        y5.backward(torch.ones_like(y5))
    ops.rng.seed(torch.device(DEV), 0x5EED)


def test_dropout_placement_matches_reference_sites():
    """Where masks are drawn (reference: embedding dropout 0.1 hierarchy_net.py:40,50; TCN Dropout(0.3) after each of the
    two ReLUs of every TemporalBlock tcn.py:23,29; nn.GRU inter-layer dropout on the outputs of layers 0..L-2 only
    hierarchy_net.py:88,213) and that eval() draws none.  Counted through the Philox call-site counter."""
    from ha2g_amd import hierarchy_net as hn, ops
    from ha2g_amd.config import hierarchy_args
    from ha2g_testing import SpeakerVocab
    args = hierarchy_args(hidden_size=32, n_layers=2)
    dev = torch.device(DEV)
    txt = hn.TextEncoderTCN(args, 40, 300, None, dropout=args.dropout_prob).to(dev)
    tok = torch.randint(0, 40, (3, 34), device=dev)
    ops.rng.seed(dev, 5)
    ops.rng.begin_step()
    txt(tok)
    assert ops.rng.call == 1 + 2 * args.n_layers                 # embedding + two per TemporalBlock
    assert txt.drop.p == 0.1 and all(b.p == 0.3 for b in txt.tcn.network)
    ops.rng.begin_step()
    txt.eval()
    txt(tok)
    assert ops.rng.call == 0
    gru = hn.BiGRU(8, 64, 4, dropout=0.3).to(dev)
    ops.rng.begin_step()
    x = torch.randn(3, 28, 8, device=dev)
    y, _ = gru(x)
    assert ops.rng.call == 3                                       # layers 0..2, not after the last layer
    # the last layer's output carries no mask: no exact zeros, and it equals a no-dropout run fed the same masked inputs
    assert float((y == 0).float().mean()) == 0.0
    # inter-layer mask really is applied: with p -> 0 masks the result changes
    gru.dropout = 0.0
    ops.rng.begin_step()
    y0, _ = gru(x)
    assert ops.rng.call == 0 and not torch.equal(y0, y)
    g1 = hn.Hierarchical_PoseGenerator(args, 27, 40, 300, None, z_obj=SpeakerVocab(6)).to(dev)
    ops.rng.begin_step()
    pre = torch.zeros(3, 34, 28, device=dev)
    g1(pre, tok, torch.randn(3, 34, 32, device=dev), torch.randint(1, 6, (3,), device=dev))
    assert ops.rng.call == (1 + 2 * args.n_layers) + (args.n_layers - 1)
    ops.rng.seed(dev, 0x5EED)


@pytest.mark.parametrize('specname', ['gesture', 'expressive'])
def test_pre_seq_pack_and_scatter_match_reference_slice_writes(specname):
    """ha2g_pre_seq_* (csrc/pack.hip) against the reference's literal construction -- zeros, `pre[:, :n, :-1] = target`,
    `pre[:, :n, -1] = 1`, then the level's slice assignments `pre[:, n:, dst] = out[:, n:, src]` in order (incl. the
    expressive off-by-one head scatter that overwrites the constraint-bit column) -- forward bit for bit, and the gradient
    w.r.t. the coarser level's output against torch autograd of those slice writes."""
    from ha2g_amd import ops
    from ha2g_amd.config import EXPRESSIVE_SPEC, GESTURE_SPEC
    spec = GESTURE_SPEC if specname == 'gesture' else EXPRESSIVE_SPEC
    dims, n, R, T = spec['pose_dims'], 4, 5, 34
    for k, P in enumerate(dims):
        Pprev = dims[k - 1] if k else 0
        tgt = rnd((R, T, P), 50 + k)
        prev = rnd((R, T, Pprev), 60 + k) if k else None
        w = rnd((R, T, P + 1), 70 + k)
        # reference construction on CPU (train_hierarchy.py:153-169 / train_hierarchy_expressive.py:163-212)
        pv = prev.clone().requires_grad_(True) if k else None
        pre = tgt.new_zeros((R, T, P + 1))
        pre[:, 0:n, :-1] = tgt[:, 0:n]
        pre[:, 0:n, -1] = 1
        for dst, src in spec['scatter'][k]:
            pre[:, n:, dst] = pv[:, n:, src]
        tables = ops.scatter_tables(P, Pprev, spec['scatter'][k], torch.device(DEV))
        pg = prev.to(DEV).requires_grad_(True) if k else None
        got = ops.pre_seq(tgt.to(DEV), pg, tables, n)
        assert torch.equal(got.detach().cpu(), pre.detach()), (specname, k)
        if k:
            (pre * w).sum().backward()
            (got * w.to(DEV)).sum().backward()
            assert torch.equal(pg.grad.cpu(), pv.grad), (specname, k)


def test_gen_concat_matches_torch_cat():
    from ha2g_amd import ops
    R, T = 7, 34
    for Wa in (16, 28, 127):
        a, b, c, z = rnd((R, T, Wa), 1), rnd((R, T, 32), 2), rnd((R, T, 32), 3), rnd((R, 16), 4)
        w = rnd((R, T, Wa + 80), 5)
        cpu = [t.clone().requires_grad_(True) for t in (a, b, c, z)]
        ref = torch.cat((cpu[0], cpu[1], cpu[2], cpu[3].unsqueeze(1).expand(-1, T, -1)), dim=2)      # hierarchy_net.py:121-141
        (ref * w).sum().backward()
        gpu = [t.to(DEV).requires_grad_(True) for t in (a, b, c, z)]
        got = ops.gen_concat(*gpu)
        assert torch.equal(got.detach().cpu(), ref.detach())
        (got * w.to(DEV)).sum().backward()
        for g_, c_ in zip(gpu[:3], cpu[:3]):
            assert torch.equal(g_.grad.cpu(), c_.grad)
        assert relerr(gpu[3].grad, cpu[3].grad) < 1e-6                 # sum over 34 frames: order of additions may differ


def test_weighted_sum_loss_assembly():
    from ha2g_amd import ops
    vals = [0.31, 1.7, -2.5, 0.004, 12.0]
    ws = [70.0, 0.1, 0.05, -0.005, 0.01]
    ts = [torch.tensor(v, device=DEV, requires_grad=True) for v in vals]
    tot = ops.weighted_sum(ts, ws)
    ref = np.float32(0)
    for v, w_ in zip(vals, ws):
        ref = np.float32(ref + np.float32(w_) * np.float32(v))
    assert abs(float(tot) - float(ref)) <= 1e-6 * abs(float(ref))
    (3.0 * tot).backward()
    for t, w_ in zip(ts, ws):
        assert abs(float(t.grad) - 3.0 * w_) <= 1e-6 * abs(3.0 * w_)


@pytest.mark.gpu
def test_multi_weight_norm_equals_per_layer_launches():
    """ha2g_weight_norm_multi_{fwd,bwd}_f32: many same-shape weight-norm layers in one launch == the per-layer kernels, bit for bit (weights and
    both parameter gradients, returned and accumulated)."""
    from ha2g_amd import ops
    n = 35                                                               # crosses the 32-per-launch chunk
    gen = torch.Generator(device=DEV).manual_seed(3)
    gs = [torch.rand(40, 1, 1, device=DEV, generator=gen).add_(0.5).requires_grad_(True) for _ in range(n)]
    vs = [torch.randn(40, 24, 2, device=DEV, generator=gen).requires_grad_(True) for _ in range(n)]
    cot = [torch.randn(40, 24, 2, device=DEV, generator=gen) for _ in range(n)]
    w_multi = ops.weight_norm_multi(gs, vs)
    g_multi = torch.autograd.grad(sum((w * c).sum() for w, c in zip(w_multi, cot)), gs + vs)
    w_one = [ops.weight_norm(g, v) for g, v in zip(gs, vs)]
    g_one = torch.autograd.grad(sum((w * c).sum() for w, c in zip(w_one, cot)), gs + vs)
    for a, b in zip(w_multi, w_one):
        assert torch.equal(a, b)
    for a, b in zip(g_multi, g_one):
        assert torch.equal(a, b)
    ref = vs[0].double() * (gs[0].double() / vs[0].double().flatten(1).norm(dim=1).view(-1, 1, 1))
    assert relerr(w_multi[0].detach(), ref.detach().cpu()) < 1e-6


@pytest.mark.parametrize('N,C', [(128, 32), (5, 64), (17, 128), (128, 256)])
def test_se_mlp_backward_data_path_fused(N, C):
    """ha2g_se_mlp_bwd_f32 (one launch) vs float64 and vs the three-launch form it replaces (GEMM, ReLU', GEMM): the data path of the SE
    excitation MLP's backward (ResNetBlocks.py:84-89 under autograd), reduction 8."""
    from ha2g_amd import ops, wav_engine as we
    torch.manual_seed(7)
    R, HW = C // 8, 144
    dsc = torch.randn(N, C, device=DEV)
    h1 = torch.relu(torch.randn(N, R, device=DEV))
    w2, w0 = torch.randn(C, R, device=DEV) * 0.2, torch.randn(R, C, device=DEV) * 0.2
    dh1, dpool = we.se_mlp_bwd(dsc, h1, w2, w0, HW)
    ref_dh1 = (dsc.double() @ w2.double()) * (h1.double() > 0)
    ref_dpool = (ref_dh1 @ w0.double()) / HW
    assert relerr(dh1, ref_dh1.cpu()) < 2e-6 and relerr(dpool, ref_dpool.cpu()) < 2e-6
    old = we.SE_MLP_FUSED
    we.SE_MLP_FUSED = False
    try:
        dh1_3, dpool_3 = we.se_mlp_bwd(dsc, h1, w2, w0, HW)
    finally:
        we.SE_MLP_FUSED = old
    # the replaced form runs its two products on the split-bf16 inner product in the default mode (4e-6 rms per GEMM); the fused kernel is plain fp32
    assert relerr(dh1, dh1_3.double().cpu()) < 5e-5 and relerr(dpool, dpool_3.double().cpu()) < 5e-5
    assert torch.equal(dh1 == 0, (h1 <= 0) | (dh1 == 0))                        # masked where the hidden unit was inactive


@pytest.mark.gpu
@pytest.mark.parametrize('N,C,HW,gpi', [(5, 64, 2240, 20), (3, 128, 576, 4), (7, 256, 144, 1)])
def test_se_excitation_mlp_forward_in_one_launch(N, C, HW, gpi):
    """ha2g_se_mlp_fwd_f32 (squeeze -> fc.0 -> ReLU -> fc.2 -> sigmoid, model/ResNetBlocks.py:84-89) against float64, from a given squeeze and from
    per-tile column sums of bn2's input (tiles inside one image, gpi per image)."""
    from ha2g_amd._lib import check, lib
    from ha2g_amd.ops import _stream
    torch.manual_seed(N * C)
    R = C // 8
    dev = DEV
    w0, b0 = torch.randn(R, C, device=dev) * C ** -0.5, torch.randn(R, device=dev) * 0.1
    w2, b2 = torch.randn(C, R, device=dev) * R ** -0.5, torch.randn(C, device=dev) * 0.1
    part = torch.randn(2, C, N * gpi, dtype=torch.float64, device=dev) * HW / gpi
    mean, invstd = torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    sums = part[0].view(C, N, gpi).sum(2).t()                                              # [N, C]
    pooled64 = (sums / HW - mean.double()) * invstd.double() * gamma.double() + beta.double()
    h64 = torch.relu(pooled64 @ w0.double().t() + b0.double())
    s64 = torch.sigmoid(h64 @ w2.double().t() + b2.double())
    pooled, h1, sc = torch.empty(N, C, device=dev), torch.empty(N, R, device=dev), torch.empty(N, C, device=dev)
    st = _stream()
    check(lib.ha2g_se_mlp_fwd_f32(None, part.data_ptr(), N * gpi, HW, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), w0.data_ptr(),
                                  b0.data_ptr(), w2.data_ptr(), b2.data_ptr(), pooled.data_ptr(), h1.data_ptr(), sc.data_ptr(), N, C, R, st))
    assert float((pooled.double() - pooled64).abs().max()) < 1e-6 * float(pooled64.abs().max())
    assert float((h1.double() - h64).abs().max()) < 2e-6 * float(h64.abs().max() + 1)
    assert float((sc.double() - s64).abs().max()) < 2e-6
    h1b, scb = torch.empty_like(h1), torch.empty_like(sc)
    check(lib.ha2g_se_mlp_fwd_f32(pooled.data_ptr(), None, 0, 0, None, None, None, None, w0.data_ptr(), b0.data_ptr(), w2.data_ptr(), b2.data_ptr(), None,
                                  h1b.data_ptr(), scb.data_ptr(), N, C, R, st))
    assert torch.equal(h1b, h1) and torch.equal(scb, sc)
    pooled_b = torch.empty_like(pooled)
    check(lib.ha2g_bn_pool_from_partials_f32(part.data_ptr(), N * gpi, N, HW, C, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                             pooled_b.data_ptr(), st))
    assert torch.equal(pooled_b, pooled)


@pytest.mark.gpu
@pytest.mark.parametrize('uses', [1, 2])
def test_weight_norm_convolution_gradients_with_deferred_joins(uses):
    """Inside the train step (SideStream.allow_defer) a convolution backward may hand the dW of a weight-normalised weight to the weight-norm
    backward WITHOUT making the main stream wait: both run on the side stream -- but only when the weight has exactly ONE gradient-carrying use
    (ops._count_grad_use): with two, autograd adds the two dW tensors on the main stream, so the functions must join.  Both cases against torch."""
    from ha2g_amd import ops
    dev = torch.device(DEV)
    torch.manual_seed(uses)
    g = torch.nn.Parameter(torch.rand(48, 1, 1, device=dev) + 0.5)
    v = torch.nn.Parameter(torch.randn(48, 32, 2, device=dev))
    g.grad, v.grad = torch.zeros_like(g), torch.zeros_like(v)              # installed buffers: the in-place path
    xs = [torch.randn(64, 34, 32, device=dev) for _ in range(uses)]
    prev, ops.side.allow_defer = ops.side.allow_defer, True
    try:
        w = ops.weight_norm(g, v)
        ys = [ops.conv1d_tm(x, w, None, dil=2, pad_left=2, To=34) for x in xs]
        assert getattr(w, '_ha2g_grad_uses', 0) == uses
        sum((y * y).sum() for y in ys).backward()
        pending = bool(ops.side._deferred.get((dev.type, dev.index)))
        ops.side.flush(dev)
    finally:
        ops.side.allow_defer = prev
    assert pending                                                          # the weight-norm backward itself defers (in-place targets)
    torch.cuda.synchronize()
    g2, v2 = g.detach().clone().requires_grad_(True), v.detach().clone().requires_grad_(True)
    w2 = g2 * v2 / v2.flatten(1).norm(dim=1).view(-1, 1, 1)
    tot = 0
    for x in xs:
        xp = torch.nn.functional.pad(x.transpose(1, 2), (2, 0))
        y = torch.nn.functional.conv1d(xp, w2, dilation=2).transpose(1, 2)
        tot = tot + (y * y).sum()
    tot.backward()
    assert float((g.grad - g2.grad).abs().max()) < 2e-4 * float(g2.grad.abs().max())
    assert float((v.grad - v2.grad).abs().max()) < 2e-4 * float(v2.grad.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize('grouped', [False, True])
def test_tcn_block_dropouts_fused_with_their_neighbours_change_no_bit(grouped):
    """Round 6: the two dropouts of a TemporalBlock (model/tcn.py:21-31) ride on their neighbours -- conv1's backward shares one launch with conv1's ReLU'
    (ha2g_dropout_fused_f32 mode 2), conv2's forward is applied by the residual add + ReLU (mode 1) and its backward by conv2's -- and draw the masks the
    stand-alone dropout launches drew (same call ids, same element indices): output and every gradient are BIT-IDENTICAL to the unfused block
    (ops.FUSE_TCN_DROPOUT = False), for the per-module block and for the generators' grouped encoders."""
    from ha2g_amd import hierarchy_net as hn, ops
    from ha2g_amd.config import hierarchy_args
    dev = torch.device(DEV)
    torch.manual_seed(4)
    if not grouped:
        blk = hn.TemporalBlock(300, 300, 2, 1, 4, 4, dropout=0.3).to(dev).train()
        x0 = torch.randn(9, 34, 300, device=dev)
        params = list(blk.parameters())

        def run():
            x = x0.clone().requires_grad_(True)
            y = blk(x)
            g = torch.autograd.grad((y * y).sum(), [x] + params)
            return [y.detach()] + [t.detach() for t in g]
    else:
        args = hierarchy_args()
        encs = [hn.TextEncoderTCN(args, 50, 300, dropout=0.3).to(dev).train() for _ in range(3)]
        tok = torch.randint(0, 50, (7, 34), device=dev)
        params = [p for e in encs for p in e.parameters()]

        def run():
            y = hn.grouped_text_encoders(encs, tok)
            g = torch.autograd.grad((y * y).sum(), params, allow_unused=True)
            return [y.detach()] + [t.detach() for t in g if t is not None]
    outs = {}
    # (neighbour fusion, conv1's dropout applied by conv2's im2col -- ha2g_im2col1d_drop_f32, the second half of round 6)
    for fuse in ((True, True), (True, False), (False, False)):
        old = ops.FUSE_TCN_DROPOUT, ops.FUSE_IM2COL_DROPOUT
        ops.FUSE_TCN_DROPOUT, ops.FUSE_IM2COL_DROPOUT = fuse
        try:
            ops.rng.seed(dev, 77)
            ops.rng.begin_step()
            outs[fuse] = run()
            torch.cuda.synchronize()
        finally:
            ops.FUSE_TCN_DROPOUT, ops.FUSE_IM2COL_DROPOUT = old
    ref = outs[(False, False)]
    assert len(ref) > 5
    assert float(ref[0].abs().max()) > 0
    if not grouped:
        assert float((ref[0] == 0).float().mean()) > 0.2                                        # the block's output: ReLU really zeroes things
    for fuse in ((True, True), (True, False)):
        assert len(outs[fuse]) == len(ref)
        for a, b in zip(outs[fuse], ref):
            assert torch.equal(a, b), fuse


@pytest.mark.gpu
@pytest.mark.parametrize('grad_slice', [None, (32, 16)])
def test_gru_inter_layer_dropout_without_a_mask_tensor_changes_no_bit(grad_slice):
    """Round 6: nn.GRU's inter-layer dropout (model/hierarchy_net.py:87) as a spec (ops.gru_drop_spec) -- dropout(y) in one launch per layer instead of a
    mask launch + a multiply, the backward re-draws the mask over the gradient-carrying row slice (ha2g_dropout_slice_f32) -- against the mask-tensor form
    (ops.GRU_MASK_SPEC = False): same call ids, same element indices, so output and every gradient are BIT-IDENTICAL; with and without the fused chains'
    row slice."""
    from ha2g_amd import hierarchy_net as hn, ops
    dev = torch.device(DEV)
    torch.manual_seed(11)
    gru = hn.BiGRU(48, 300, 3, dropout=0.3).to(dev).train()
    x0 = torch.randn(48, 34, 48, device=dev)
    params = list(gru.parameters())
    outs = {}
    for spec in (True, False):
        old = ops.GRU_MASK_SPEC
        ops.GRU_MASK_SPEC = spec
        gru.grad_slice = grad_slice
        try:
            ops.rng.seed(dev, 5)
            ops.rng.begin_step()
            x = x0.clone().requires_grad_(True)
            y, _ = gru(x)
            g = torch.autograd.grad((y * y).sum(), [x] + params)
            outs[spec] = [y.detach()] + [t.detach() for t in g]
            torch.cuda.synchronize()
        finally:
            ops.GRU_MASK_SPEC = old
            gru.grad_slice = None
    assert ops.gru_cluster_error(dev) == 0
    for a, b in zip(outs[True], outs[False]):
        assert torch.equal(a, b)
    assert float(outs[True][1].abs().max()) > 0


@pytest.mark.gpu
def test_gru_stack_weight_operands_split_in_one_launch_change_no_bit():
    """Round 6: the merged input-projection operands [W_ih; W_ih_reverse] of layers 1 .. L-1 of an nn.GRU stack (model/hierarchy_net.py:87) are split into piece
    planes by ONE launch for the whole stack (forward: [6H][K] planes; backward, for dX: their transposes, both directions side by side) instead of a
    torch.cat + a split per layer: the same plane values, so the output and every gradient are BIT-IDENTICAL to the per-layer form (ops.GRU_STACK_PREP = False).
    64 x 34 rows: the plane GEMM serves the projections."""
    from ha2g_amd import hierarchy_net as hn, ops
    dev = torch.device(DEV)
    torch.manual_seed(3)
    gru = hn.BiGRU(107, 300, 4, dropout=0.0).to(dev).train()
    x0 = torch.randn(64, 34, 107, device=dev)
    params = list(gru.parameters())
    outs = {}
    for prep in (True, False):
        old = ops.GRU_STACK_PREP
        ops.GRU_STACK_PREP = prep
        try:
            x = x0.clone().requires_grad_(True)
            y, _ = gru(x)
            g = torch.autograd.grad((y * torch.sin(y)).sum(), [x] + params)
            outs[prep] = [y.detach()] + [t.detach() for t in g]
            torch.cuda.synchronize()
        finally:
            ops.GRU_STACK_PREP = old
    assert ops.gru_cluster_error(dev) == 0
    for a, b in zip(outs[True], outs[False]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('rows,cols,ld', [(300000, 16, 16), (123457, 32, 32), (70001, 64, 64), (9000, 64, 80), (5000, 300, 300), (4096, 256, 256), (700, 64, 64), (4099, 1024, 1024)])
def test_column_sums_narrow_and_wide_forms_against_float64(rows, cols, ld):
    """ha2g_colsum_f32 (bias gradients; model/ResNetSE34V2.py:157-185's tap convolutions have 16 / 32 / 64 output channels): the 16-byte-load form for narrow
    contiguous matrices (round 6) and the 64-lane form, both against a float64 sum at 1e-6 of sum |x|; beta accumulates; repeated calls are bit-identical."""
    from ha2g_amd import ops
    g = torch.Generator().manual_seed(rows + cols)
    buf = (torch.randn(rows, ld, generator=g) + 0.3).to(DEV)
    x = buf[:, :cols]
    ref = x.double().sum(0)
    scale = x.double().abs().sum(0)
    got = ops.colsum(x)
    assert float(((got.double() - ref).abs() / scale).max()) < 1e-6
    assert torch.equal(got, ops.colsum(x))
    acc = torch.full((cols,), 2.0, device=DEV)
    ops.colsum(x, out=acc, beta=0.5)
    assert float(((acc.double() - (ref + 1.0)).abs() / (scale + 1.0)).max()) < 1e-6

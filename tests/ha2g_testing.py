"""Shared parity-test helpers: procedural state/batches and the golden-fixture tolerance policy.

Tolerance policy (DESIGN.md "Parity"): an fp32 implementation passes on array X when
    max|X - truth64| <= 1e-4 * max|truth64| + 3 * max(noise32(X), cond(X)) + tiny,
all three taken from runs of the REFERENCE stored in tests/golden: truth64 = its float64 result; noise32 = the largest
deviation from it over four float32 runs of the reference (the plain one and three with inputs perturbed by one float32
ulp, which re-rolls every rounding decision downstream); cond = how far its float64 result moves under that one-ulp
input perturbation.  On outputs, losses, generator / text / discriminator gradients these floors are ~1e-7..1e-6, i.e. the
1e-4 term rules.  On the SE-ResNet's gradients at B = 3..4 the reference's own fp32 runs scatter by ~1e-3 (train-mode
BatchNorm over 3-4 samples makes the backward chaotic), and that measured scatter is the floor.  For gradient tensors
the floor is never taken below the median relative floor of the same module's gradients.  float64 must hit 1e-9.
"""
import re

import numpy as np
import torch

from ha2g_amd import procedural as proc
from ha2g_amd import schema
from ha2g_amd.config import FGD_CASE


def state_for(case, dt=torch.float32, dims=schema.GESTURE_POSE_DIMS):
    sch = schema.step_schema(dims, case['n_words'], case['n_spk'], case['hidden_size'], case['n_layers'])
    sd = schema.procedural_state(sch, case['seed'])
    return {k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd.items()}


def batch_for(case, dt=torch.float32, P=27):
    text, spec, target, vid = map(torch.from_numpy,
                                  proc.make_batch(case['B'], P, case['n_words'], case['n_spk'], case['seed']))
    return text, spec.to(dt), target.to(dt), vid


def wproc(name, t, seed):
    w = proc.tensor_for('w.' + name, (2,) + tuple(t.shape), seed)[0] * t[0].numel() ** 0.5
    return torch.from_numpy(w).to(device=t.device, dtype=t.dtype)


def leaf_params(sd, role):
    ps = {k: v for k, v in sd.items() if k.startswith(role + '.') and v.is_floating_point()
          and not k.endswith(('running_mean', 'running_var')) and '.net.' not in k}
    for v in ps.values():
        v.requires_grad_(True)
    return ps


def _np(t):
    if isinstance(t, torch.Tensor):
        t = t.detach().double().cpu().numpy()
    return np.asarray(t, np.float64)


class Checker:
    margins = []   # (error / tolerance, key) of every comparison made, for tools/margins.py

    def __init__(self, g, dt=torch.float32, rtol=1e-4, noise_mult=3.0, tail=None):
        """tail: the 200-run tail study of the same case (tests/golden/cfg1_tail.npz, gen_tail_study.py) -- for the tensors it covers (the audio
        tower's gradients of step 0) the reference's own fp32 scatter is the maximum over those 200 runs (heavy-tailed: up to 5x the maximum
        over the fixture's 25), entering the tolerance as in tests/test_gpu_tail.py: 1e-4 * scale + 1.5 * max over the 200 reference runs."""
        self.g = g
        self.tail = tail
        self.f64 = dt == torch.float64
        self.rtol = 1e-9 if self.f64 else rtol
        self.nm = 0.0 if self.f64 else noise_mult
        self.worst = 0.0
        self._grp = {}

    def _group_of(self, key):
        """'step0/grad/audio.feat_extractor.x' -> 'step0/grad/audio.' ; 'audio/grad/feat_extractor.x' -> 'audio/grad/'"""
        parts = key.split('/')
        if parts[0].startswith('step') and len(parts) >= 3:
            return '/'.join(parts[:2]) + '/' + parts[2].split('.')[0] + '.'
        if len(parts) >= 3 and parts[1] == 'grad':
            return '/'.join(parts[:2]) + '/'
        return None

    def _group_rel_floor(self, key):
        """Median relative floor over all gradient tensors of the same module: the conditioning level of that
        module's backward pass at this test point (a single tensor's own three-draw floor has a heavy lower tail)."""
        return self._group_stats(key)[0]

    def _group_stats(self, key):
        """(median relative floor, median magnitude) over the gradient tensors of the same module."""
        grp = self._group_of(key)
        if grp is None:
            return 0.0, 0.0
        if grp not in self._grp:
            rel, mag = [], []
            for k in self.g.files:
                if k.startswith(grp) and k.endswith('/sample'):
                    s = np.abs(self.g[k]).max()
                    if s > 0:
                        rel.append(self._floor(k) / s)
                        mag.append(s)
            self._grp[grp] = (float(np.median(rel)), float(np.median(mag))) if rel else (0.0, 0.0)
        return self._grp[grp]

    def _floor(self, key):
        """The reference's own fp32 scatter on this array, or its ulp-conditioning floor, whichever is larger.  For a tensor of one of the
        generators g1..g6 the scatter is pooled over the same-named tensor of ALL the generators (relative to each one's scale): they are one
        architecture on same-scale data, and what makes the scatter heavy-tailed is a rare discrete event -- one LeakyReLU / ReLU output of the
        head or the text encoder within rounding of zero flips its derivative (1 <-> 0.01) and moves every gradient of THAT generator by up to
        1e-3 of its scale (tools/diag_g2_event.py, profiles/r04_kink_event_cfg3_b128.txt: exact-fp32 runs of this framework that differ by
        one-ulp input moves land on either side in 4 of 12 draws).  Each generator's 25+ reference runs are draws from the same distribution."""
        own = self._floor1(key)
        m = re.match(r'^(.*/)g(\d+)\.(.*)$', key)
        if m is None:
            return own
        scale = self._scale_of(key)
        pooled = 0.0
        for i in range(1, 9):
            k2 = '%sg%d.%s' % (m.group(1), i, m.group(3))
            if k2 + '@noise' in self.g.files:
                pooled = max(pooled, self._floor1(k2) / self._scale_of(k2))
        return max(own, pooled * scale)

    def _scale_of(self, key):
        return max(float(np.abs(self.g[key]).max()), 1e-30)

    def _floor1(self, key):
        n = float(self.g[key + '@noise'])
        ck = key + '@cond'
        n = max(n, float(self.g[ck])) if ck in self.g.files else n
        if self.tail is not None and self.nm > 0 and key + '@dev' in self.tail.files:
            n = max(n, 1.5 / self.nm * float(self.tail[key + '@dev'].max()))
        return n

    def _tol(self, key, scale):
        return self.rtol * scale + self.nm * self._floor(key) + 1e-12 + (0 if self.f64 else 1e-7 * scale)

    def close(self, got, key):
        ref = self.g[key]
        got = _np(got).reshape(ref.shape)
        scale = max(np.abs(ref).max(), 1e-30)
        err = np.abs(got - ref).max()
        tol = self._tol(key, scale)
        self.worst = max(self.worst, err / scale)
        Checker.margins.append((float(err / tol), key))
        assert err <= tol, '%s: max err %.3e > tol %.3e (scale %.3e, ref fp32 noise %.3e)' % (
            key, err, tol, scale, float(self.g[key + '@noise']))

    def digest(self, got, key, norms_only=False):
        """Compare (norm, strided sample) digests of a large tensor."""
        a = _np(got).reshape(-1)
        stride = max(1, a.size // 64)
        smp = a[::stride][:64]
        nrm = np.sqrt((a * a).sum())
        rn = float(self.g[key + '/norm'])
        # | ||g+d|| - ||g|| | <= ||d|| <= sqrt(numel) * max|d_i|: the element-wise floor bounds the norm's floor too
        # (a norm's own noise sample can be accidentally tiny when d happens to be orthogonal to g).
        gfl = self._group_rel_floor(key + '/sample')
        tol = self._tol(key + '/norm', max(rn, 1e-30)) + self.nm * self._floor(key + '/sample') * np.sqrt(a.size) + self.nm * gfl * rn
        Checker.margins.append((float(abs(nrm - rn) / tol), key + '/norm'))
        assert abs(nrm - rn) <= tol, '%s/norm: %.9e vs %.9e (tol %.2e)' % (key, nrm, rn, tol)
        ref = self.g[key + '/sample']
        # elementwise: scale by the tensor's rms-ish magnitude so tiny sampled entries are not over-weighted
        scale = max(np.abs(ref).max(), rn / max(np.sqrt(a.size), 1.0), 1e-30)
        err = np.abs(smp - ref).max()
        # + rtol of the module's typical gradient magnitude: a tensor whose gradient is itself the residue of heavy
        # cancellation (e.g. an SE gate's  sum_hw dout*bn(x)  ~1e-3 of its terms) is judged on the module's scale.
        tol = self._tol(key + '/sample', scale) + self.nm * gfl * scale + (0 if self.f64 else self.rtol * self._group_stats(key + '/sample')[1])
        Checker.margins.append((float(err / tol), key + '/sample'))
        assert err <= tol, '%s/sample: max err %.3e > tol %.3e (scale %.3e)' % (key, err, tol, scale)

    def grads(self, prefix, role, params, grads):
        for (k, _), gr in zip(params.items(), grads):
            self.digest(gr, '%s/%s' % (prefix, k[len(role) + 1:]))

    def step(self, si, ret, grads, sd, lr=5e-4):
        g = self.g
        pre = 'step%d/' % si
        if si > 0 and not self.f64:
            # The 2nd step runs on Adam-updated weights: sign flips of noise-level gradients feed back into
            # every tensor, so two fp32 runs of the REFERENCE itself scatter at the 1e-3 level here (B=3..4).
            # The exact state logic (Adam moments, D-phase ordering, grad accumulation on D) is pinned by the
            # float64 run; in fp32 only the loss dict and gradient norms are checked, loosely.
            relaxed = Checker(g, torch.float32, rtol=2e-3, noise_mult=self.nm * 4)
            return relaxed._step(si, ret, grads, sd, lr, norms_only=True)
        return self._step(si, ret, grads, sd, lr)

    def _step(self, si, ret, grads, sd, lr, norms_only=False):
        g = self.g
        pre = 'step%d/' % si
        ref_keys = sorted(k[len(pre) + 4:] for k in g.files if k.startswith(pre + 'ret/') and '@' not in k)
        assert sorted(ret) == ref_keys, (sorted(ret), ref_keys)
        for k, v in ret.items():
            r = float(g[pre + 'ret/' + k])
            tol = self._tol(pre + 'ret/' + k, max(abs(r), 1e-3))
            assert abs(v - r) <= tol, (si, k, v, r, tol)
        for k in g.files:
            if '@' in k or not k.startswith(pre) or not k.endswith('/norm'):
                continue
            kind, pk = k[len(pre):-5].split('/', 1)
            if '.net.' in pk:
                continue
            if norms_only:
                if kind == 'grad':
                    nrm = float(np.sqrt((_np(grads[pk]) ** 2).sum()))
                    rn = float(g[k])
                    assert abs(nrm - rn) <= 2e-2 * rn + 4 * self._floor(k) + 1e-9, (si, pk, nrm, rn)
                continue
            if kind == 'grad':
                self.digest(grads[pk], k[:-5])
            elif kind == 'buf':
                self.digest(sd[pk], k[:-5])
            elif kind == 'param':
                if self.f64:
                    self.digest(sd[pk], k[:-5])
                else:
                    # Adam's first step moves every weight by lr * sign(g): where |g| is inside the reference's own
                    # rounding noise the sign is arbitrary, so those elements may differ by 2*lr; all others must
                    # match within 2 % of the step size.
                    a = _np(sd[pk]).reshape(-1)
                    smp = a[::max(1, a.size // 64)][:64]
                    err = np.abs(smp - g[k[:-5] + '/sample'])
                    gk = pre + 'grad/' + pk
                    if gk + '/sample' in g.files:
                        gs = np.abs(g[gk + '/sample'])
                        sure = gs > 8 * self._floor(gk + '/sample') + 1e-4 * max(gs.max(), 1e-30)
                    else:
                        sure = np.ones_like(err, bool)
                    tol = 0.02 * lr * (si + 1) + 4 * self._floor(k[:-5] + '/sample')
                    assert (err[sure] <= tol).all(), (si, pk, float(err[sure].max()))
                    assert (err <= 2.05 * lr * (si + 1) + tol).all(), (si, pk, float(err.max()))


# ---- module construction helpers shared by GPU tests, smoke() and bench.py -------------------------------

class SpeakerVocab:
    """Stand-in for the reference's vocab.Vocab used as z_obj: .n_words is read on the hot path, .word2index by the validation loop's
    random speaker draw (scripts/train.py:365)."""

    def __init__(self, n_words):
        self.n_words = n_words
        self.word2index = {'spk%d' % i: i for i in range(n_words)}


def no_dropout(m):
    from ha2g_amd.hierarchy_net import BiGRU, TemporalBlock
    for sub in m.modules():
        if isinstance(sub, torch.nn.Dropout):
            sub.p = 0.0
        if isinstance(sub, BiGRU):
            sub.dropout = 0.0
        if isinstance(sub, TemporalBlock):
            sub.p = 0.0
    return m


def load_role(module, state, role):
    """Load the `role.`-prefixed entries of a step state dict into one module."""
    sub = {k[len(role) + 1:]: v for k, v in state.items() if k.startswith(role + '.')}
    module.load_state_dict(sub)
    return module


def build_modules(case, device, dims=schema.GESTURE_POSE_DIMS, state=None, keep_dropout=False):
    """(args, [g1..], dis, audio, text) with procedural parameters of `case`, dropout disabled (keep_dropout: left as the modules define it)."""
    global no_dropout
    _nd = no_dropout
    if keep_dropout:
        no_dropout = lambda m: m
    try:
        return _build_modules(case, device, dims, state, 0.3 if keep_dropout else None)
    finally:
        no_dropout = _nd


def _build_modules(case, device, dims, state, dropout_prob=None):
    from ha2g_amd.config import make_args
    from ha2g_amd import hierarchy_net as hn
    args = make_args(case)
    if dropout_prob is not None:
        args.dropout_prob = dropout_prob                     # config/hierarchy.yml's value (the parity cases build with 0)
    spk = SpeakerVocab(case['n_spk'])
    state = state if state is not None else state_for(case, torch.float32, dims)
    gens = []
    for i, pd in enumerate(dims):
        g = hn.Hierarchical_PoseGenerator(args, pd, case['n_words'], 300, None, z_obj=spk)
        gens.append(no_dropout(load_role(g, state, 'g%d' % (i + 1))).to(device))
    dis = no_dropout(load_role(hn.Hierarchical_ConvDiscriminator(dims[-1]), state, 'dis')).to(device)
    aud = no_dropout(load_role(hn.Hierarchical_WavEncoder(args, spk, len(dims), 32), state, 'audio')).to(device)
    txt = no_dropout(load_role(hn.TextEncoderTCN(args, case['n_words'], 300, None, dropout=args.dropout_prob), state, 'text')).to(device)
    return args, gens, dis, aud, txt


class EpsInjector:
    """Feeds the reference's reparameterisation-noise sequence to the generators.  The reference draws one (B,16)
    tensor per generator call in the order [chain][g1,g2,g3]; the fused step calls each generator once with k*B rows
    (row block p = chain eps_block_order[p]), so generator j receives stream entries base + 3*i + j for i = 0..k-1."""

    def __init__(self, gens, seed, B):
        self.B, self.base, self.seed = B, 0, seed
        self.n_gen = len(gens)
        self.gens = list(gens)
        for j, g in enumerate(gens):
            g.eps_source = (lambda shape, device, j=j: self(j, shape, device))

    def _draw(self, k):
        r = np.random.Generator(np.random.PCG64([self.seed, 991, k]))
        return r.standard_normal((self.B, 16)).astype(np.float32)

    def __call__(self, j, shape, device):
        k = shape[0] // self.B
        # physical row block p holds the reference's pass number order[p] (the step puts the gradient-carrying block last)
        order = getattr(self.gens[j], 'eps_block_order', None) or list(range(k))
        eps = np.concatenate([self._draw(self.base + self.n_gen * order[p] + j) for p in range(k)], 0)
        if j == self.n_gen - 1:
            self.base += self.n_gen * k
        return torch.from_numpy(eps).to(device)


def named_state(mods):
    """{'g1.xxx': tensor} over parameters and buffers of a {role: module} dict; grads likewise."""
    sd, grads = {}, {}
    for role, m in mods.items():
        for k, p in m.named_parameters():
            sd['%s.%s' % (role, k)] = p
            if p.grad is not None:
                grads['%s.%s' % (role, k)] = p.grad
        for k, b in m.named_buffers():
            sd['%s.%s' % (role, k)] = b
    return sd, grads


# ---- round 2: audio-tower fixtures (tests/golden/blocks.npz, blocksfull.npz, enc16.npz) -------------------------------------

class DigestChecker:
    """Checks (norm, strided sample) digests written by gen_golden.digest_n:
        |x - truth| <= rtol * scale + noise_mult * max(@noise, @cond)      (float64: 1e-9 * scale)
    `shares` records, per comparison, which fraction of the tolerance the reference's own fp32 scatter contributes --
    on the strict fixtures (blocks.npz) it must stay below 10 %, i.e. rtol = 1e-4 is the operative bound."""

    def __init__(self, g, dt=torch.float32, rtol=1e-4, noise_mult=1.0):
        self.g = g
        self.f64 = dt == torch.float64
        self.rtol = 1e-9 if self.f64 else rtol
        self.nm = 0.0 if self.f64 else noise_mult
        self.shares, self.errs = [], []

    def floor(self, key):
        return max(float(self.g[key + '@noise']), float(self.g[key + '@cond']))

    def check(self, got, key):
        g = self.g
        a = _np(got).reshape(-1)
        ref = g[key + '/sample']
        smp = a if a.size <= 4096 else a[::max(1, a.size // ref.size)][:ref.size]
        assert smp.shape == ref.shape, (key, smp.shape, ref.shape)
        rn = float(g[key + '/norm'])
        nrm = float(np.sqrt((a * a).sum()))
        ntol = self.rtol * max(rn, 1e-30) + self.nm * self.floor(key + '/norm') + self.nm * self.floor(key + '/sample') * np.sqrt(a.size)
        assert abs(nrm - rn) <= ntol, '%s/norm: %.9e vs %.9e (tol %.2e)' % (key, nrm, rn, ntol)
        scale = max(np.abs(ref).max(), rn / max(np.sqrt(a.size), 1.0), 1e-30)
        fl = self.nm * self.floor(key + '/sample')
        tol = self.rtol * scale + fl + 1e-12
        err = np.abs(smp - ref).max()
        self.shares.append((fl / tol, key))
        self.errs.append((err / scale, key))
        assert err <= tol, '%s/sample: max err %.3e > tol %.3e (scale %.3e, rel %.2e)' % (key, err, tol, scale, err / scale)


def block_state(name, geom, seed, dt=torch.float32):
    """Procedural state dict {key: tensor} of one SEBasicBlock fixture (keys as the reference registers them)."""
    cin, c, h, w, first = geom
    sch = schema.se_block_schema(cin, c, first, 'blk.%s.' % name)
    sd = schema.procedural_state(sch, seed)
    return {k[len('blk.%s.' % name):]: (v.to(dt) if v.is_floating_point() else v) for k, v in sd.items()}


def block_io(name, geom, B, seed, dt=torch.float32):
    """(x NCHW, loss weights NCHW for out) of a block fixture."""
    cin, c, h, w, first = geom
    x = torch.from_numpy(proc.block_input('blk.%s.x' % name, (B, cin, h, w), seed)).to(dt)
    oh, ow = ((h + 1) // 2, (w + 1) // 2) if first else (h, w)
    wl = torch.from_numpy(proc.tensor_for('w.blk.%s' % name, (2, B, c, oh, ow), seed)[0] * (c * oh * ow) ** 0.5).to(dt)
    return x, wl


def engine_P(sd, device):
    """name -> tensor / wav_engine._BN on `device` from a flat state dict with the reference's key names
    (BatchNorm entries 'x.weight/.bias/.running_mean/.running_var/.num_batches_tracked' fold into one _BN under 'x')."""
    from ha2g_amd.wav_engine import _BN
    P = {}
    for k, v in sd.items():
        if k.endswith('.running_mean'):
            q = k[:-len('running_mean')]
            P[q[:-1]] = _BN(*(sd[q + n].clone().to(device) for n in ('weight', 'bias', 'running_mean', 'running_var', 'num_batches_tracked')))
    for k, v in sd.items():
        base = k.rsplit('.', 1)[0]
        if base in P:
            continue
        t = v.clone().to(device)
        if t.dim() == 4:
            t = t.contiguous(memory_format=torch.channels_last)
        P[k] = t
    return P


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2)


def taps_inputs(case, seed, dt=torch.float32):
    B, W3 = case['B'], case['W3']
    shapes = dict(layer2=(B, 64, 64, 4 * W3 - 1), layer3=(B, 128, 32, 2 * W3), layer4=(B, 256, 16, W3))
    feats = {k: torch.from_numpy(proc.block_input('taps.' + k, shp, seed)).to(dt) for k, shp in shapes.items()}
    vid = torch.from_numpy(np.arange(B, dtype=np.int64) % (case['n_spk'] - 1) + 1)
    return feats, vid


def taps_w(name, t, seed):
    return torch.from_numpy(proc.tensor_for('w.taps.' + name, (2,) + tuple(t.shape), seed)[0] * t[0].numel() ** 0.5).to(device=t.device, dtype=t.dtype)


def tcn_relu_margin(case):
    """Smallest non-zero |ReLU input| (relative to the tensor's max) over the text encoders of a step case, from the
    procedural state in float64 (torch CPU conv1d; test-side helper).  The token inputs are integers, so the reference's
    nine perturbed fp32 runs never re-roll these decisions: a pre-activation within fp32 rounding of zero would be an
    un-measured coin flip of any fp32 implementation.  Step fixtures are generated from seeds where this margin is safe."""
    import torch.nn.functional as F
    dims = schema.EXPRESSIVE_POSE_DIMS if case.get('expressive') else schema.GESTURE_POSE_DIMS
    sd = state_for(case, torch.float64, dims)
    text = torch.from_numpy(proc.make_batch(case['B'], dims[-1], case['n_words'], case['n_spk'], case['seed'])[0])
    worst = 1.0
    for role in ['g%d.text_encoder.' % (i + 1) for i in range(len(dims))] + ['text.']:
        x = F.embedding(text, sd[role + 'embedding.weight']).transpose(1, 2)
        for i in range(case['n_layers']):
            dil, y = 2 ** i, x
            for c in ('conv1', 'conv2'):
                q = '%stcn.network.%d.%s.' % (role, i, c)
                v = sd[q + 'weight_v']
                w = sd[q + 'weight_g'] * v / v.flatten(1).norm(dim=1).view(-1, 1, 1)
                y = F.conv1d(F.pad(y, (dil, 0)), w, sd[q + 'bias'], dilation=dil)
                a = y.abs()
                worst = min(worst, float(a[a > 0].min() / a.max()))
                y = torch.relu(y)
            dk = '%stcn.network.%d.downsample.weight' % (role, i)
            res = F.conv1d(x, sd[dk], sd[dk[:-6] + 'bias']) if dk in sd else x
            s = (y + res).abs()
            worst = min(worst, float(s[s > 0].min() / s.max()))
            x = torch.relu(y + res)
    return worst


def fgd_ae_state(dt=torch.float32, P=27):
    """Procedural state dict of the reference's EmbeddingNet(mode='pose') (P=27; keys/shapes restated from model/embedding_net.py:42-82,
    156-218) or MotionAE(126, 128) (P=126; model/motion_ae.py:33-131)."""
    sch = {}
    latent = 32 if P == 27 else 128

    def bn(p, c):
        for k in ('weight', 'bias', 'running_mean', 'running_var'):
            sch[p + k] = (c,)
        sch[p + 'num_batches_tracked'] = ()
    e, d = ('pose_encoder.' if P == 27 else 'encoder.'), 'decoder.'
    for i, (co, ci, k) in enumerate(((32, P, 3), (64, 32, 3), (64, 64, 4))):
        sch['%snet.%d.0.weight' % (e, i)] = (co, ci, k); sch['%snet.%d.0.bias' % (e, i)] = (co,); bn('%snet.%d.1.' % (e, i), co)
    sch[e + 'net.3.weight'] = (32, 64, 3); sch[e + 'net.3.bias'] = (32,)
    for i, (co, ci) in ((0, (256, 384)), (3, (128, 256)), (6, (latent, 128))):
        sch['%sout_net.%d.weight' % (e, i)] = (co, ci); sch['%sout_net.%d.bias' % (e, i)] = (co,)
    bn(e + 'out_net.1.', 256); bn(e + 'out_net.4.', 128)
    if P == 27:
        for k in ('fc_mu', 'fc_logvar'):
            sch[e + k + '.weight'] = (32, 32); sch[e + k + '.bias'] = (32,)
    sch[d + 'pre_net.0.weight'] = (64, latent); sch[d + 'pre_net.0.bias'] = (64,); bn(d + 'pre_net.1.', 64)
    sch[d + 'pre_net.3.weight'] = (136, 64); sch[d + 'pre_net.3.bias'] = (136,)
    sch[d + 'net.0.weight'] = (4, 32, 3); sch[d + 'net.0.bias'] = (32,); bn(d + 'net.1.', 32)
    sch[d + 'net.3.weight'] = (32, 32, 3); sch[d + 'net.3.bias'] = (32,); bn(d + 'net.4.', 32)
    sch[d + 'net.6.weight'] = (32, 32, 3); sch[d + 'net.6.bias'] = (32,)
    sch[d + 'net.7.weight'] = (P, 32, 3); sch[d + 'net.7.bias'] = (P,)
    sd = {}
    for k, shp in sch.items():
        v = proc.tensor_for(('fgd.' if P == 27 else 'fgd126.') + k, shp if shp else (1,), FGD_CASE['seed'])
        t = torch.from_numpy(np.asarray(v)).reshape(shp)
        sd[k] = t.to(dt) if t.is_floating_point() else t
    return sd


def tolerance_profile(g, prefix='step0/grad/', rtol=1e-4, tail=None):
    """What Checker.digest will ALLOW on every gradient digest under `prefix` of fixture `g`, as a fraction of the tensor's scale -- a property of
    the fixture and the tolerance policy alone (no implementation involved).  -> {module: (share of comparisons whose relative tolerance exceeds
    2e-4, median relative tolerance, count)} plus the same under 'all'.  tests/test_tolerance_freeze.py pins these numbers: any change to the
    Checker that loosens a comparison shows up as a grown share and fails (the checker is frozen)."""
    ck = Checker(g, tail=tail, rtol=rtol)
    per = {}
    for k in g.files:
        if not (k.startswith(prefix) and k.endswith('/sample')) or '.net.' in k:
            continue
        key = k[:-7]
        ref = g[k]
        rn = float(g[key + '/norm'])
        n_el = max(ref.size, 1)
        # Checker.digest's elementwise branch (the norm of the full tensor is not stored with its element count: use the sample's own scale rule)
        scale = max(np.abs(ref).max(), 1e-30)
        gfl = ck._group_rel_floor(k)
        tol = ck._tol(k, scale) + ck.nm * gfl * scale + ck.rtol * ck._group_stats(k)[1]
        mod = key[len(prefix):].split('.')[0]
        per.setdefault(mod, []).append(tol / scale)
        per.setdefault('all', []).append(tol / scale)
    return {m: (float(np.mean(np.asarray(v) > 2e-4)), float(np.median(v)), len(v)) for m, v in per.items()}

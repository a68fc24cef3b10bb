"""CPU ORACLE for the HA2G hierarchy train step -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch functional restatement (torch CPU fp32 primitives over plain dicts of tensors) of the
reference algorithm on the hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import it; nothing under ha2g_amd/ does.

Parity status: PINNED.  Every function here is checked in tests/test_oracle_golden.py against the
fixtures tests/golden/*.npz, which were produced by importing and running the reference itself
(tests/golden/gen_golden.py) -- module outputs, gradients, two full train steps (warm-up phase and
GAN phase) incl. Adam updates and BatchNorm running statistics, Gesture and Expressive variants.

State is a flat dict  name -> tensor  using the reference's state_dict keys
prefixed by the module's role: 'g1.'..'g6.', 'dis.', 'audio.', 'text.'.

Reference citations are relative to /root/reference/scripts/.
"""
import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------------------------


def batch_norm_train(x, sd, p, momentum=0.1, eps=1e-5, update=True):
    """Train-mode BatchNorm over all dims but channel (dim 1): biased batch variance to normalise,
    unbiased variance into running_var, momentum 0.1 (torch semantics).  Uses the
    fused torch primitive so that the float32 backward has the same (analytic) form as the reference's."""
    if update == 'eval':                                             # inference: normalise with the running statistics
        return F.batch_norm(x, sd[p + 'running_mean'], sd[p + 'running_var'], sd[p + 'weight'], sd[p + 'bias'], False, momentum, eps)
    rm = sd[p + 'running_mean'] if update else None
    rv = sd[p + 'running_var'] if update else None
    if update:
        sd[p + 'num_batches_tracked'].add_(1)
    return F.batch_norm(x, rm, rv, sd[p + 'weight'], sd[p + 'bias'], True, momentum, eps)


_DROP_SEQ = [None]                  # iterator over pre-scaled dropout masks in the reference's call order, or None (drop_sequence)


class drop_sequence:
    """`with drop_sequence(masks=[m0, m1, ...]):` -- dropout ON with GIVEN masks: every dropout of the text encoders (embedding, the two per TCN block:
    model/tcn.py:21-31), of nn.GRU between layers (model/hierarchy_net.py:87, :217) consumes the next mask, in the order the reference calls them
    (time-major [B, T, C] masks are transposed to [B, C, T] at the TCN sites).  tests/test_gpu_linearised.py feeds the masks the HIP step drew."""

    def __init__(self, masks):
        self.new = iter(masks)

    def __enter__(self):
        self.old = _DROP_SEQ[0]
        _DROP_SEQ[0] = self.new
        return self

    def __exit__(self, *exc):
        it = _DROP_SEQ[0]
        _DROP_SEQ[0] = self.old
        if exc[0] is None:
            left = sum(1 for _ in it)
            assert left == 0, 'drop_sequence: %d masks were not consumed (the call order of the two sides differs)' % left


def _next_drop(x, time_major_mask=False):
    """x times the next mask of the sequence (None active: x unchanged)"""
    if _DROP_SEQ[0] is None:
        return x
    m = next(_DROP_SEQ[0])
    if time_major_mask:
        m = m.transpose(1, 2)
    assert tuple(m.shape) == tuple(x.shape), ('drop_sequence: mask of shape %s at a site of shape %s' % (tuple(m.shape), tuple(x.shape)))
    return x * m.to(x.dtype)


def gru_bidir(x, sd, p, n_layers, H, masks=None):
    """Stacked bidirectional GRU, batch_first, h0 = 0 (torch.nn.GRU semantics).
    gates ordered (r,z,n); n = tanh(gi_n + r*(W_hn h + b_hn)); h' = (1-z)*n + z*h.
    `masks[l]` (already scaled by 1/(1-p)) multiplies the output of layer l < last (dropout)."""
    B, T, _ = x.shape
    inp = x
    for l in range(n_layers):
        outs = []
        for suffix, order in (('', range(T)), ('_reverse', range(T - 1, -1, -1))):
            w_ih, w_hh = sd['%sweight_ih_l%d%s' % (p, l, suffix)], sd['%sweight_hh_l%d%s' % (p, l, suffix)]
            b_ih, b_hh = sd['%sbias_ih_l%d%s' % (p, l, suffix)], sd['%sbias_hh_l%d%s' % (p, l, suffix)]
            gi_all = inp @ w_ih.t() + b_ih
            h = x.new_zeros(B, H)
            ys = [None] * T
            for t in order:
                gi = gi_all[:, t]
                gh = h @ w_hh.t() + b_hh
                r = torch.sigmoid(gi[:, :H] + gh[:, :H])
                z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
                n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
                h = (1 - z) * n + z * h
                ys[t] = h
            outs.append(torch.stack(ys, 1))
        inp = torch.cat(outs, 2)
        if masks is not None and l < n_layers - 1:
            inp = inp * masks[l]
        elif l < n_layers - 1:
            inp = _next_drop(inp)
    return inp


def weight_norm(g, v):
    """w = g * v / ||v||, norm over (in, k) per output channel (tcn.py:19 via torch weight_norm)."""
    return g * v / v.flatten(1).norm(dim=1).view(-1, 1, 1)


# ----------------------------------------------------------------------------------------------
# modules
# ----------------------------------------------------------------------------------------------


def text_encoder_tcn(tokens, sd, p, n_layers, drop=None):
    """model/hierarchy_net.py:48-52 + model/tcn.py:16-64: embedding -> n_layers x [2 x (causal dilated
    conv k=2 -> ReLU)] + identity residual -> ReLU -> Linear(->32).  `drop` = optional dict of
    pre-scaled dropout masks {'emb', (i,0), (i,1)}."""
    x = F.embedding(tokens, sd[p + 'embedding.weight'])
    if drop:
        x = x * drop['emb']
    else:
        x = _next_drop(x)
    x = x.transpose(1, 2)                                            # (B, C, T)
    for i in range(n_layers):
        dil = 2 ** i
        res = x
        y = x
        for j, c in enumerate(('conv1', 'conv2')):
            q = '%stcn.network.%d.%s.' % (p, i, c)
            w = weight_norm(sd[q + 'weight_g'], sd[q + 'weight_v'])
            y = F.conv1d(F.pad(y, (dil, 0)), w, sd[q + 'bias'], dilation=dil)   # left pad == pad+chomp
            y = _act(y, 0.0, True)
            if drop:
                y = y * drop[(i, j)]
            else:
                y = _next_drop(y, True)
        if (p + 'tcn.network.%d.downsample.weight' % i) in sd:
            res = F.conv1d(res, sd[p + 'tcn.network.%d.downsample.weight' % i], sd[p + 'tcn.network.%d.downsample.bias' % i])
        x = _act(y + res, 0.0, True)
    return F.linear(x.transpose(1, 2), sd[p + 'decoder.weight'], sd[p + 'decoder.bias'])


class _StoreBf16(torch.autograd.Function):
    """A tensor that the bf16-storage mode keeps in HBM as bf16: the forward value is rounded to bf16 (nearest even) where it is stored,
    and so is the gradient that arrives at it (the activation-gradient stream is stored as bf16 as well); straight-through otherwise."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _GradBf16(torch.autograd.Function):
    """Identity forward; the gradient arriving here is rounded to bf16 (a point where the activation-gradient stream is stored)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _WeightBf16(torch.autograd.Function):
    """A convolution weight as the bf16 matrix kernels read it -- and any tensor stored rounded whose incoming gradient is NOT rounded at
    this point: rounded forward, gradient untouched (weight gradients are fp32)."""

    @staticmethod
    def forward(ctx, w):
        return w.to(torch.bfloat16).to(w.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


_BF16_STORAGE = [False]


class bf16_storage:
    """`with bf16_storage():` -- evaluate the audio encoder's trunk with the rounding points of ha2g_amd's bf16-storage mode (BASELINE config 5;
    the reference has no such mode: this is the oracle *under the stated storage format*, tests/test_gpu_b16.py).  Rounded: the stem output and
    its BatchNorm, the five stored tensors of each SEBasicBlock (ReLU(conv1), bn1, conv2, bn2, block output), the downsample conv and its
    BatchNorm, every 3x3 / 1x1 trunk convolution weight, and the activation gradients where ha2g_amd/wav_b16.py stores them (one rounding
    per stored gradient tensor, in its order of accumulation).  Not rounded: statistics, the SE squeeze (taken from the unrounded bn2
    output) and gate, parameter gradients, the taps and everything after them."""

    def __enter__(self):
        self.prev = _BF16_STORAGE[0]
        _BF16_STORAGE[0] = True

    def __exit__(self, *a):
        _BF16_STORAGE[0] = self.prev


_RELU_SITES = [None, None]          # [masks to impose (site -> bool tensor) or None, dict to record the oracle's own decisions into or None]


class relu_pattern:
    """`with relu_pattern(masks=m):` -- evaluate the audio encoder with every ReLU decision IMPOSED: relu(x) becomes x * m[site] at the
    sites named below.  That is the reference's network linearised at a given activation pattern: where the pattern is the one an fp32
    implementation's forward actually took, its backward can be held to the float64 result at 1e-4 without the ReLU-flip scatter that
    two fp32 runs of the reference itself show (tests/test_gpu_linearised.py).  `record={}` collects the oracle's own decisions (x > 0),
    so that imposing them reproduces the plain oracle exactly (tests/test_oracle_blocks.py).  Sites (tensors NCHW / [N, R]):
    'stem', '<block>c1' (ReLU(conv1), ResNetBlocks.py:24-25), '<block>se' (SELayer's hidden ReLU, :86), '<block>out' (the block's
    final ReLU, :36), 'tap_low' / 'tap_mid' / 'tap_high' (ResNetSE34V2.py:158,168,179)."""

    def __init__(self, masks=None, record=None):
        self.new = [masks, record]

    def __enter__(self):
        self.prev = list(_RELU_SITES)
        _RELU_SITES[:] = self.new

    def __exit__(self, *a):
        _RELU_SITES[:] = self.prev


def _relu(x, site):
    masks, record = _RELU_SITES
    if record is not None:
        record[site] = (x > 0).detach()
    if masks is not None and site in masks:
        return x * masks[site].to(x.dtype)
    return torch.relu(x)


_ACT_SEQ = [None, None]             # [iterator over masks to impose, list to record into] for the activations OUTSIDE the audio tower, in call order


class act_sequence:
    """`with act_sequence(masks=[m0, m1, ...]):` -- the ReLU / LeakyReLU decisions of the text encoders (model/tcn.py:21-29,43-46), the generators' head
    (model/hierarchy_net.py:91) and the discriminator (model/hierarchy_net.py:200-206) imposed in CALL ORDER (the k-th such activation evaluated takes
    the k-th mask; time-major [B, T, C] masks are transposed to the oracle's [B, C, T] where the site says so): relu(x) -> x * m, leaky(x) -> x * (m + 0.01 (1 - m)).
    With relu_pattern (the tower's sites, by name) this linearises the WHOLE train step at a given activation pattern.  `record=[]` collects the
    oracle's own decisions in the same order and layout convention."""

    def __init__(self, masks=None, record=None):
        self.new = [iter(masks) if masks is not None else None, record]

    def __enter__(self):
        self.prev = list(_ACT_SEQ)
        _ACT_SEQ[:] = self.new

    def __exit__(self, *a):
        left = 0 if _ACT_SEQ[0] is None else sum(1 for _ in _ACT_SEQ[0])
        _ACT_SEQ[:] = self.prev
        assert left == 0, 'act_sequence: %d masks were not consumed (the call order of the two sides differs)' % left


def _act(x, slope, time_major_mask):
    """relu (slope 0) / leaky_relu at one sequence site; x in the oracle's layout, the mask in the other side's when time_major_mask"""
    it, record = _ACT_SEQ
    if record is not None:
        m = (x > 0).detach()
        record.append(m.transpose(1, 2).contiguous() if time_major_mask else m)
    if it is not None:
        m = next(it)
        if time_major_mask:
            m = m.reshape(x.shape[0], x.shape[2], x.shape[1]).transpose(1, 2)
        m = m.reshape(x.shape).to(x.dtype)
        return x * (m + slope * (1 - m)) if slope else x * m
    return F.leaky_relu(x, slope) if slope else torch.relu(x)


def _st(x):
    return _StoreBf16.apply(x) if _BF16_STORAGE[0] else x


def _wq(w):
    return _WeightBf16.apply(w) if _BF16_STORAGE[0] else w


def _gr(x):
    return _GradBf16.apply(x) if _BF16_STORAGE[0] else x


def se_block(x, sd, p, stride, has_down, update_bn=True):
    """model/ResNetBlocks.py:21-37: conv -> ReLU -> BN -> conv -> BN -> SE -> (+residual) -> ReLU.  (_st / _wq are identities unless the
    bf16-storage rounding points are switched on, see bf16_storage.)"""
    # bf16 storage: a first block's conv1 data gradient is stored (rounded) before the downsample branch's is added onto it
    out = F.conv2d(_gr(x) if has_down else x, _wq(sd[p + 'conv1.weight']), None, stride=stride, padding=1)
    out = _st(batch_norm_train(_st(_relu(out, p + 'c1')), sd, p + 'bn1.', update=update_bn))
    out = _st(F.conv2d(out, _wq(sd[p + 'conv2.weight']), None, padding=1))
    # bf16 storage: bn2's output is stored rounded, the squeeze is taken from the unrounded values of the same pass, and the gradient w.r.t.
    # bn2's output (gate path + squeeze path) is rounded once, where it is stored
    out = _gr(batch_norm_train(out, sd, p + 'bn2.', update=update_bn))
    y = out.mean((2, 3))                                             # ResNetBlocks.py:91-95
    out = _wq(out)
    y = _relu(F.linear(y, sd[p + 'se.fc.0.weight'], sd[p + 'se.fc.0.bias']), p + 'se')
    y = torch.sigmoid(F.linear(y, sd[p + 'se.fc.2.weight'], sd[p + 'se.fc.2.bias']))
    out = out * y[:, :, None, None]
    if has_down:
        x = _st(F.conv2d(x, _wq(sd[p + 'downsample.0.weight']), None, stride=stride))
        x = _st(batch_norm_train(x, sd, p + 'downsample.1.', update=update_bn))
    return _st(_relu(out + x, p + 'out'))


def wav_encoder(spec, vid, sd, p, pose_level, update_bn=True):
    """model/hierarchy_net.py:16-19 -> model/ResNetSE34V2.py:118-218.  Returns
    (weight (B,3,L), feat_low, feat_mid, feat_high (B,T,32), [blend_i (B,T,32)] * L)."""
    q = p + 'feat_extractor.'
    B = spec.shape[0]
    x = F.conv2d(spec.unsqueeze(1), sd[q + 'conv1.weight'], sd[q + 'conv1.bias'], padding=1)
    x = _st(batch_norm_train(_st(_relu(x, 'stem')), sd, q + 'bn1.', update=update_bn))
    feats = []
    for li, nblk in enumerate((3, 4, 6, 3)):
        for j in range(nblk):
            first = j == 0 and li > 0
            x = se_block(x, sd, '%slayer%d.%d.' % (q, li + 1, j), 2 if first else 1, first, update_bn)
        feats.append(x)
        x = _gr(x)                 # bf16 storage: the next layer's data gradient is stored before the tap's gradient is added onto it

    low = wav_tap(feats[1], sd, q, 'low', 1, update_bn)
    mid = wav_tap(feats[2], sd, q, 'mid', 2, update_bn)
    high = wav_tap(feats[3], sd, q, 'high', 4, update_bn)
    w, blend = wav_blend(vid, low, mid, high, sd, q, pose_level)
    return w, low, mid, high, blend


def wav_tap(f, sd, q, name, shuffle, update_bn=True):
    """One tap of ResNetSE.forward (model/ResNetSE34V2.py:157-192): [PixelShuffle] -> conv -> ReLU -> BN ->
    (B, C*H, W) -> transpose -> Linear; the FC's K index is c*H + h, its row index the time column w."""
    B = f.shape[0]
    if shuffle > 1:
        f = F.pixel_shuffle(f, shuffle)
    f = F.conv2d(f, sd[q + 'conv_%s.weight' % name], sd[q + 'conv_%s.bias' % name])
    f = batch_norm_train(_relu(f, 'tap_' + name), sd, q + 'bn_%s.' % name, update=update_bn)
    f = f.reshape(B, -1, f.shape[-1]).transpose(1, 2)            # (B, W, C*H)
    return F.linear(f, sd[q + 'fc_%s.weight' % name], sd[q + 'fc_%s.bias' % name])


def wav_blend(vid, low, mid, high, sd, q, pose_level):
    """Speaker-conditioned softmax blending of the three taps (model/ResNetSE34V2.py:194-214)."""
    B = low.shape[0]
    z = F.embedding(vid, sd[q + 'speaker_embedding.0.weight'])
    z = F.linear(z, sd[q + 'speaker_embedding.1.weight'], sd[q + 'speaker_embedding.1.bias'])
    h = F.elu(F.linear(F.elu(z), sd[q + 'fc1.weight'], sd[q + 'fc1.bias']))
    w = F.linear(h, sd[q + 'fc2.weight'], sd[q + 'fc2.bias']).reshape(B, 3, pose_level).softmax(1)
    blend = [low * w[:, 0, i, None, None] + mid * w[:, 1, i, None, None] + high * w[:, 2, i, None, None]
             for i in range(pose_level)]
    return w, blend


def pose_generator(pre_seq, tokens, audio_feat, vid, sd, p, n_layers, H, eps, drop=None):
    """model/hierarchy_net.py:99-149.  eps: (B,16) reparameterisation noise.  drop: optional dict with
    'text' (text-encoder masks) and 'gru' (list of layer masks)."""
    text = text_encoder_tcn(tokens, sd, p + 'text_encoder.', n_layers, drop['text'] if drop else None)
    z = F.embedding(vid, sd[p + 'speaker_embedding.0.weight'])
    z = F.linear(z, sd[p + 'speaker_embedding.1.weight'], sd[p + 'speaker_embedding.1.bias'])
    mu = F.linear(z, sd[p + 'speaker_mu.weight'], sd[p + 'speaker_mu.bias'])
    logvar = F.linear(z, sd[p + 'speaker_logvar.weight'], sd[p + 'speaker_logvar.bias'])
    zc = mu + eps * torch.exp(0.5 * logvar)                          # embedding_net.py:10-13
    x = torch.cat((pre_seq, audio_feat, text, zc.unsqueeze(1).expand(-1, pre_seq.shape[1], -1)), 2)
    y = gru_bidir(x, sd, p + 'gru.', n_layers, H, drop['gru'] if drop else None)
    y = y[:, :, :H] + y[:, :, H:]
    y = F.linear(y, sd[p + 'out.0.weight'], sd[p + 'out.0.bias'])
    y = _act(y, 0.01, False)
    y = F.linear(y, sd[p + 'out.2.weight'], sd[p + 'out.2.bias'])
    return y, zc, mu, logvar


def conv_discriminator(poses, sd, p, update_bn=True, gru_masks=None):
    """model/hierarchy_net.py:222-242: 3 x Conv1d(k=3) (+BN+LeakyReLU on the first two) -> bi-GRU(8->64,
    4 layers) -> direction sum -> Linear(64,1) per frame -> Linear(T-6,1) -> sigmoid."""
    x = poses.transpose(1, 2)
    x = F.conv1d(x, sd[p + 'pre_conv.0.weight'], sd[p + 'pre_conv.0.bias'])
    x = _act(batch_norm_train(x, sd, p + 'pre_conv.1.', update=update_bn), 0.01, True)
    x = F.conv1d(x, sd[p + 'pre_conv.3.weight'], sd[p + 'pre_conv.3.bias'])
    x = _act(batch_norm_train(x, sd, p + 'pre_conv.4.', update=update_bn), 0.01, True)
    x = F.conv1d(x, sd[p + 'pre_conv.6.weight'], sd[p + 'pre_conv.6.bias']).transpose(1, 2)
    y = gru_bidir(x, sd, p + 'gru.', 4, 64, gru_masks)
    y = y[:, :, :64] + y[:, :, 64:]
    y = F.linear(y, sd[p + 'out.weight'], sd[p + 'out.bias']).squeeze(2)
    return torch.sigmoid(F.linear(y, sd[p + 'out2.weight'], sd[p + 'out2.bias']))


# ----------------------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------------------


def contrastive_ce_rows(a, b, s, e, expressive=False):
    """Rows s..e-1 of the loss below, already divided by N: sum_i CE(logits[i, :], i) / N (so that row blocks can be
    back-propagated one at a time at N = 8704, where the whole N x N x 32 graph would not fit in memory)."""
    an = F.normalize(a, p=2, dim=1)
    bn = F.normalize(b, p=2, dim=1)
    d = (an[s:e, None, :] - bn[None, :, :]).norm(p=2, dim=2)
    logits = 1.0 / d if expressive else torch.clamp(1.0 / (d + 1e-8), min=1e-8)
    return F.cross_entropy(logits, torch.arange(s, e), reduction='sum') / a.shape[0]


def contrastive_ce(a, b, expressive=False, chunk=1024):
    """train_eval/train_hierarchy.py:54-68 (Gesture: +1e-8, clamp) and
    train_hierarchy_expressive.py:107-121 (no eps, no clamp).  dist[i,j] = ||a_i - b_j||, rows = a.
    Computed in row chunks (same arithmetic; avoids the N x N x 32 intermediate of the reference)."""
    N = a.shape[0]
    total = a.new_zeros(())
    for s in range(0, N, chunk):
        total = total + contrastive_ce_rows(a, b, s, min(s + chunk, N), expressive)
    return total


def huber(x, y, beta, reduction='mean'):
    """smooth_l1_loss(x/beta, y/beta) * beta == Huber(delta=beta) (train_hierarchy.py:173-176)."""
    d = (x - y).abs()
    v = torch.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta)
    return v.mean() if reduction == 'mean' else v


def physical_prior(out, mean_dir_vec, pairs, avg, var, palm=()):
    """train_hierarchy.py:242-262 / train_hierarchy_expressive.py:421-447.  `palm`: bone pairs whose raw cross products
    are appended as extra "bones" before the normalisation (left / right palm normals)."""
    raw = out + mean_dir_vec.view(1, 1, -1)
    v = raw.reshape(raw.shape[0] * raw.shape[1], -1, 3)
    if palm:
        extra = [torch.cross(v[:, a], v[:, b], dim=1).unsqueeze(1) for a, b in palm]
        v = torch.cat([v] + extra, 1)
    v = F.normalize(v, dim=-1)
    total = 0
    for i, (a, b) in enumerate(pairs):
        ip = torch.clamp((v[:, a] * v[:, b]).sum(1), -1 + 1e-7, 1 - 1e-7)
        ang = torch.acos(ip) / math.pi
        total = total + torch.mean((ang - avg[i]) ** 2 / (2 * var[i]))
    return total


# ----------------------------------------------------------------------------------------------
# train step.  The hierarchy tables (level columns, scatter maps, angle statistics) are data restated from the
# reference in ha2g_amd/config.py::{GESTURE,EXPRESSIVE}_SPEC.
# ----------------------------------------------------------------------------------------------


def _adam(params, grads, state, lr, step, b1=0.5, b2=0.999, eps=1e-8):
    """torch.optim.Adam semantics (no weight decay, no amsgrad): train.py:155-170."""
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    with torch.no_grad():
        for k, p in params.items():
            g = grads.get(k)
            if g is None:
                continue
            m, v = state.setdefault(k, (torch.zeros_like(p), torch.zeros_like(p)))
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            p.addcdiv_(m, (v.sqrt() / math.sqrt(bc2)).add_(eps), value=-lr / bc1)


class OracleTrainer:
    """Holds the state dict, Adam states and step counters of the six (nine) modules and runs
    train_iter_hierarchy (train_eval/train_hierarchy.py:71-293) on CPU."""

    def __init__(self, sd, args, spec=None):
        from ha2g_amd.config import GESTURE_SPEC
        self.spec = spec or GESTURE_SPEC
        self.sd = sd
        self.args = args
        self.L = n_levels = len(self.spec['pose_dims'])
        self.roles = ['g%d' % (i + 1) for i in range(n_levels)] + ['dis', 'audio', 'text']
        self.adam = {r: {} for r in self.roles}
        self.steps = {r: 0 for r in self.roles}
        self.grads = {}                                              # name -> accumulated .grad (dis keeps across phases)

    def params(self, role):
        return {k: v for k, v in self.sd.items() if k.startswith(role + '.') and v.is_floating_point()
                and not k.endswith(('running_mean', 'running_var')) and '.net.' not in k}

    def _chain(self, tokens, target_lv, blend, vid, eps_fn, grad):
        a = self.args
        outs = []
        last = None
        ctx = torch.enable_grad() if grad else torch.no_grad()
        with ctx:
            for k in range(self.L):
                tk = target_lv[k]
                pre = tk.new_zeros(tk.shape[0], tk.shape[1], tk.shape[2] + 1)
                pre[:, :a.n_pre_poses, :-1] = tk[:, :a.n_pre_poses]
                pre[:, :a.n_pre_poses, -1] = 1
                for dst, src in self.spec['scatter'][k]:             # coarse output -> finer pre_seq, differentiable
                    pre[:, a.n_pre_poses:, dst] = outs[-1][:, a.n_pre_poses:, src]
                o, z, mu, lv = pose_generator(pre, tokens, blend[k], vid, self.sd, 'g%d.' % (k + 1), a.n_layers,
                                              a.hidden_size, eps_fn((tk.shape[0], 16)))
                outs.append(o)
                last = (z, mu, lv)
        return outs, last

    def train_iter(self, epoch, tokens, spec, target, vid, eps_fn, rand_perm):
        a = self.args
        sd = self.sd
        for r in self.roles:
            for k, p in self.params(r).items():
                p.requires_grad_(True)
        sp = self.spec
        _, low, mid, high, blend = wav_encoder(spec, vid, sd, 'audio.', self.L)
        text_feat = text_encoder_tcn(tokens, sd, 'text.', a.n_layers)
        tl = [target[:, :, c] for c in sp['level_cols']]
        ret = {}
        gan = epoch > a.loss_warmup and a.loss_gan_weight > 0.0
        dis_error = None
        if gan:
            for k in self.params('dis'):
                self.grads.pop(k, None)
            outs, _ = self._chain(tokens, tl, [b.detach() for b in blend], vid, eps_fn, grad=False)
            real = conv_discriminator(target, sd, 'dis.')
            fake = conv_discriminator(outs[-1].detach(), sd, 'dis.')
            dis_error = -torch.mean(torch.log(real + 1e-8) + torch.log(1 - fake + 1e-8))
            dp = self.params('dis')
            gs = torch.autograd.grad(dis_error, list(dp.values()), allow_unused=True)
            for (k, _), g in zip(dp.items(), gs):
                if g is not None:
                    self.grads[k] = g.clone()
            self.steps['dis'] += 1
            _adam(dp, self.grads, self.adam['dis'], a.learning_rate * a.discriminator_lr_weight, self.steps['dis'])
        for r in self.roles:
            if r != 'dis':
                for k in self.params(r):
                    self.grads.pop(k, None)
        N = text_feat.shape[0] * text_feat.shape[1]
        c_pos = contrastive_ce(text_feat.reshape(N, -1), high.reshape(N, -1), sp['contrastive_expressive'])
        c_neg = -contrastive_ce(text_feat.reshape(N, -1), low.reshape(N, -1), sp['contrastive_expressive'])
        outs, (z, mu, logvar) = self._chain(tokens, tl, blend, vid, eps_fn, grad=True)
        hub = sum(huber(o, t, 0.1) for o, t in zip(outs, tl))
        dis_out = conv_discriminator(outs[-1], sd, 'dis.')
        gen_error = -torch.mean(torch.log(dis_out + 1e-8))
        rvid = vid[rand_perm]
        routs, (rz, _, _) = self._chain(tokens, tl, [b.detach() for b in blend], rvid, eps_fn, grad=False)
        pose_l1 = huber(outs[-1], routs[-1].detach(), 0.05, 'none').sum((1, 2))
        z_l1 = (z.detach() - rz.detach()).abs().mean(1)
        div_reg = torch.clamp(-(pose_l1 / (z_l1 + 1e-5)), min=-1000).mean()
        kld = -0.5 * torch.mean(1 + logvar - mu.pow(2) - logvar.exp())
        loss = a.loss_regression_weight * hub + a.loss_kld_weight * kld + a.loss_reg_weight * div_reg
        if epoch > a.loss_warmup:
            loss = loss + a.loss_gan_weight * gen_error
        loss = loss + a.loss_contrastive_pos_weight * c_pos + a.loss_contrastive_neg_weight * c_neg
        mdv = torch.tensor(a.mean_dir_vec).squeeze(1)
        phy = physical_prior(outs[-1], mdv, sp['phys_pairs'], sp['phys_avg'], sp['phys_var'], sp['palm'])
        loss = loss + a.loss_physical_weight * phy
        allp = {}
        for r in self.roles:
            allp.update(self.params(r))
        gs = torch.autograd.grad(loss, list(allp.values()), allow_unused=True)
        for (k, _), g in zip(allp.items(), gs):
            if g is not None:
                self.grads[k] = self.grads[k] + g if k in self.grads else g.clone()
        for r in self.roles:
            if r == 'dis':
                continue
            self.steps[r] += 1
            _adam(self.params(r), self.grads, self.adam[r], a.learning_rate, self.steps[r])
        ret['loss'] = a.loss_regression_weight * hub.item()
        if kld.item():
            ret['KLD'] = a.loss_kld_weight * kld.item()
        if div_reg.item():
            ret['DIV_REG'] = a.loss_reg_weight * div_reg.item()
        if gan:
            ret['gen'] = a.loss_gan_weight * gen_error.item()
            ret['dis'] = dis_error.item()
        ret['c_pos'] = a.loss_contrastive_pos_weight * c_pos.item()
        ret['c_neg'] = a.loss_contrastive_neg_weight * c_neg.item()
        ret['phy'] = a.loss_physical_weight * phy.item()
        return ret


# ----------------------------------------------------------------------------------------------
# sliding-window synthesis (scripts/synthesize_hierarchy.py:36-215), hierarchy model, eval-mode modules
# ----------------------------------------------------------------------------------------------


def synthesize_windows(args, sd, spec, lang_model, n_audio, words, spectrogram, vid, eps_fn, audio_sr=16000):
    """generate_gestures_hierarchy restated over the oracle's functional modules: windows of n_poses frames every
    n_poses - n_pre_poses frames, the first n_pre_poses of a window seeded with the previous window's last ones (per level),
    overlap cross-faded prev*(n-j)/(n+1) + next*(j+1)/(n+1) in float32 numpy (:150-161).  -> numpy [frames, P]."""
    import numpy as np
    from ha2g_amd.synthesize import calc_spectrogram_length_from_motion_length, frame_tokens, plan_windows
    n_frames, n_pre = args.n_poses, args.n_pre_poses
    clip_length = n_audio / audio_sr
    unit_time, stride_time, num_sub = plan_windows(clip_length, n_frames, n_pre, args.motion_resampling_framerate)
    spec_len = calc_spectrogram_length_from_motion_length(n_frames, args.motion_resampling_framerate)
    dims = spec['pose_dims']
    dt = spectrogram.dtype
    targets = [torch.zeros(1, n_frames, P, dtype=dt) for P in dims]
    vid_t = torch.LongTensor([vid])
    out_list, out_dir_vec = [], None
    with torch.no_grad():
        for i in range(num_sub):
            start_time = i * stride_time
            a0 = math.floor(start_time / clip_length * spectrogram.shape[0])      # the reference's shape[0] (mel bins) quirk
            in_spec = spectrogram[:, a0:a0 + spec_len].unsqueeze(0)
            tokens = frame_tokens(lang_model, words, start_time, start_time + unit_time, n_frames)
            if i > 0:
                for k, c in enumerate(spec['level_cols']):
                    targets[k][:, :n_pre] = out_dir_vec[:, -n_pre:][:, :, c]
            _, _, _, _, blend = wav_encoder(in_spec, vid_t, sd, 'audio.', len(dims), update_bn='eval')
            prev = None
            for k in range(len(dims)):
                tk = targets[k]
                pre = tk.new_zeros(1, n_frames, dims[k] + 1)
                pre[:, :n_pre, :-1] = tk[:, :n_pre]
                pre[:, :n_pre, -1] = 1
                for dst, src in spec['scatter'][k]:
                    pre[:, n_pre:, dst] = prev[:, n_pre:, src]
                prev, _, _, _ = pose_generator(pre, tokens, blend[k], vid_t, sd, 'g%d.' % (k + 1), args.n_layers, args.hidden_size,
                                               eps_fn((1, 16)))
            out_dir_vec = prev
            out_seq = out_dir_vec[0].numpy().copy()
            if out_list:
                last_poses = out_list[-1][-n_pre:]
                out_list[-1] = out_list[-1][:-n_pre]
                n = len(last_poses)
                for j in range(n):
                    out_seq[j] = last_poses[j] * (n - j) / (n + 1) + out_seq[j] * (j + 1) / (n + 1)
            out_list.append(out_seq)
    return np.vstack(out_list)

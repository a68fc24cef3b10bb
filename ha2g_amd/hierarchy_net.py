"""MI355X-native mirror of the reference's model/hierarchy_net.py (+ ResNetSE34V2.py, tcn.py): same class
names, constructor arguments, forward signatures, return conventions and state_dict keys, so checkpoints
(`gen_dict_*`, `dis_dict`, `audio_dict`, `text_dict`, scripts/train.py:233-237) and call sites carry over.
Every arithmetic op runs in a hand-written HIP kernel via ha2g_amd.ops (no torch/MIOpen compute fallback).
"""
import math

import torch
import torch.nn as nn

from . import ops, wav_engine
from .ops import ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_SIGMOID


def _uniform(t, bound):
    with torch.no_grad():
        return t.uniform_(-bound, bound)


# ---------------------------------------------------------------------------------------------------
# parameter-holding leaf modules (names/shapes/initialisation as torch.nn's)
# ---------------------------------------------------------------------------------------------------

class Linear(nn.Module):
    def __init__(self, in_features, out_features):
        super().__init__()
        self.weight = nn.Parameter(_uniform(torch.empty(out_features, in_features), 1.0 / math.sqrt(in_features)))
        self.bias = nn.Parameter(_uniform(torch.empty(out_features), 1.0 / math.sqrt(in_features)))

    def forward(self, x, act=ACT_NONE):
        return ops.linear(x, self.weight, self.bias, act)


class Embedding(nn.Module):
    def __init__(self, num_embeddings, embedding_dim, _weight=None, freeze=False):
        super().__init__()
        w = torch.randn(num_embeddings, embedding_dim) if _weight is None else torch.as_tensor(_weight, dtype=torch.float32)
        self.weight = nn.Parameter(w, requires_grad=not freeze)

    def forward(self, tokens):
        return ops.embedding(tokens, self.weight)


class _Marker(nn.Module):
    """Parameter-free placeholder keeping the reference's Sequential indices (activations are fused into kernels)."""

    def forward(self, x):
        return x


class BatchNormParams(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer('running_mean', torch.zeros(c))
        self.register_buffer('running_var', torch.ones(c))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))

    def forward(self, x, act=ACT_NONE):
        """x: contiguous [..., C] (channels last)."""
        if not self.training:
            return ops.batch_norm_eval(x, self.weight, self.bias, self.running_mean, self.running_var, 1e-5, act)
        self.num_batches_tracked.add_(1)
        return ops.batch_norm_train(x, self.weight, self.bias, self.running_mean, self.running_var, 0.1, 1e-5, act)


# ---------------------------------------------------------------------------------------------------
# text encoder (reference model/hierarchy_net.py:22-52, model/tcn.py)
# ---------------------------------------------------------------------------------------------------

class WNConv1d(nn.Module):
    """weight-normalised Conv1d parameters: bias, weight_g (C,1,1), weight_v (C,Cin,k); tcn.py:19,25."""

    def __init__(self, cin, cout, k):
        super().__init__()
        v = torch.empty(cout, cin, k).normal_(0, 0.01)                       # tcn.py:38-40
        self.bias = nn.Parameter(_uniform(torch.empty(cout), 1.0 / math.sqrt(cin * k)))
        self.weight_g = nn.Parameter(v.flatten(1).norm(dim=1).view(cout, 1, 1).clone())
        self.weight_v = nn.Parameter(v)


class TemporalBlock(nn.Module):
    def __init__(self, n_inputs, n_outputs, kernel_size, stride, dilation, padding, dropout=0.2):
        super().__init__()
        self.conv1 = WNConv1d(n_inputs, n_outputs, kernel_size)
        self.conv2 = WNConv1d(n_outputs, n_outputs, kernel_size)
        # same registration pattern as the reference (each conv appears under conv{1,2} and net.{0,4})
        self.net = nn.Sequential(self.conv1, _Marker(), _Marker(), _Marker(), self.conv2, _Marker(), _Marker(), _Marker())
        self.downsample = None
        if n_inputs != n_outputs:
            self.downsample = nn.Module()
            self.downsample.weight = nn.Parameter(torch.empty(n_outputs, n_inputs, 1).normal_(0, 0.01))
            self.downsample.bias = nn.Parameter(_uniform(torch.empty(n_outputs), 1.0 / math.sqrt(n_inputs)))
        self.dilation, self.p = dilation, dropout

    def forward(self, x, ws=None):
        """x: [B, T, C] time-major (the reference carries [B, C, T]; same math).  ws: the two weight-normalised conv weights when the caller
        computed all of the network's weight norms in one launch."""
        d = self.dilation
        y = x
        spec = prev = None
        for ci, conv in enumerate((self.conv1, self.conv2)):
            w = ws[ci] if ws is not None else ops.weight_norm(conv.weight_g, conv.weight_v)
            # the block's two dropouts ride on their neighbours (ops.DropSpec): conv1's is applied by conv2's im2col (k = 2, C % 4 == 0; else by a launch of its
            # own), its backward shares one launch with conv1's ReLU'; conv2's is applied by the residual add + ReLU below
            defer = ci == 1 or ops.im2col_drop_ok(w.shape[0], self.conv2.weight_v.shape[2])
            spec = ops.make_drop(self.p, self.training, x.device, deferred=defer) if x.is_cuda else None
            y = ops.conv1d_tm(y, w, conv.bias, dil=d, pad_left=d * (w.shape[2] - 1), To=x.shape[1], act=ACT_RELU, drop=spec,
                              in_drop=prev if (prev is not None and prev.deferred) else None)
            prev = spec
            if spec is None:
                y = ops.dropout(y, self.p, self.training)
        res = x if self.downsample is None else ops.conv1d_tm(x, self.downsample.weight, self.downsample.bias)
        return ops.add_relu(y, res, drop=spec)


class TemporalConvNet(nn.Module):
    def __init__(self, num_inputs, num_channels, kernel_size=2, dropout=0.2):
        super().__init__()
        layers = []
        for i, c in enumerate(num_channels):
            layers.append(TemporalBlock(num_inputs if i == 0 else num_channels[i - 1], c, kernel_size, stride=1, dilation=2 ** i,
                                        padding=(kernel_size - 1) * 2 ** i, dropout=dropout))
        self.network = nn.Sequential(*layers)

    def forward(self, x):
        convs = [c for b in self.network for c in (b.conv1, b.conv2)]
        if x.is_cuda and len({tuple(c.weight_v.shape) for c in convs}) == 1:          # every weight norm of the stack in one launch
            ws = ops.weight_norm_multi([c.weight_g for c in convs], [c.weight_v for c in convs])
            for i, b in enumerate(self.network):
                x = b(x, ws[2 * i:2 * i + 2])
            return x
        return self.network(x)


class TextEncoderTCN(nn.Module):
    """reference model/hierarchy_net.py:22-52"""

    def __init__(self, args, n_words, embed_size=300, pre_trained_embedding=None, kernel_size=2, dropout=0.3, emb_dropout=0.1):
        super().__init__()
        if pre_trained_embedding is not None:
            assert pre_trained_embedding.shape[0] == n_words
            assert pre_trained_embedding.shape[1] == embed_size
            self.embedding = Embedding(n_words, embed_size, _weight=pre_trained_embedding, freeze=args.freeze_wordembed)
        else:
            self.embedding = Embedding(n_words, embed_size)
        num_channels = [args.hidden_size] * args.n_layers
        self.tcn = TemporalConvNet(embed_size, num_channels, kernel_size, dropout=dropout)
        self.decoder = Linear(num_channels[-1], 32)
        self.drop = nn.Dropout(emb_dropout)          # holds p only; the mask kernel is ops.dropout
        self.emb_dropout = emb_dropout
        self.init_weights()

    def init_weights(self):
        self.decoder.bias.data.fill_(0)
        self.decoder.weight.data.normal_(0, 0.01)

    def forward(self, input):
        emb = ops.dropout(self.embedding(input), self.drop.p, self.training)
        y = self.tcn(emb)                              # [B, T, C]
        return self.decoder(y).contiguous()


def text_encoders_groupable(encs):
    """True when the encoders are same-shape TextEncoderTCN modules without residual down-sampling (embed size == hidden size, the
    hierarchy.yml configuration): their layers can then run as grouped launches."""
    if len(encs) < 2 or len(encs) > 8 or not all(isinstance(e, TextEncoderTCN) for e in encs):
        return False
    e0 = encs[0]
    sig = lambda e: (tuple(e.embedding.weight.shape), len(e.tcn.network), e.drop.p, e.training,
                     tuple((tuple(b.conv1.weight_v.shape), tuple(b.conv2.weight_v.shape), b.dilation, b.p, b.downsample is None) for b in e.tcn.network),
                     tuple(e.decoder.weight.shape))
    return all(sig(e) == sig(e0) for e in encs) and all(b.downsample is None for b in e0.tcn.network)


def grouped_conv_weights(encs):
    """The weight-normalised convolution weights of every TCN layer of every encoder: [level][conv 0/1][encoder].  They depend on the
    parameters only, so a train step computes them ONCE and shares them between the no-grad and the gradient-carrying row blocks."""
    convs = [[[getattr(e.tcn.network[lvl], name) for e in encs] for name in ('conv1', 'conv2')] for lvl in range(len(encs[0].tcn.network))]
    flat = [c for lv in convs for pair in lv for c in pair]
    if len({tuple(c.weight_v.shape) for c in flat}) == 1 and flat[0].weight_v.is_cuda:          # one launch for all of them (32 per launch)
        ws = ops.weight_norm_multi([c.weight_g for c in flat], [c.weight_v for c in flat])
        it = iter(ws)
        return [[[next(it) for _ in pair] for pair in lv] for lv in convs]
    return [[[ops.weight_norm(c.weight_g, c.weight_v) for c in pair] for pair in lv] for lv in convs]


def grouped_text_encoders(encs, in_text, wn=None):
    """The G generators' text encoders (separate modules in the reference, model/hierarchy_net.py:66-70, each called inside its generator's
    forward :121-123) evaluated in lockstep: every layer is ONE launch over the stacked [G*B, T, C] activations and one grouped GEMM with the
    G weight sets, instead of G under-filled launches.  Same arithmetic per encoder; returns [G, B, T, 32].  wn: grouped_conv_weights(encs)
    when the caller evaluates several row blocks with the same parameters."""
    G = len(encs)
    B, T = in_text.shape
    e0 = encs[0]
    if wn is None:
        wn = grouped_conv_weights(encs)
    x = torch.stack([e.embedding(in_text) for e in encs])                  # [G, B, T, E]: each encoder has its own table
    E = x.shape[3]
    x = ops.dropout(x.view(G * B, T, E), e0.drop.p, e0.training)
    for lvl in range(len(e0.tcn.network)):
        blocks = [e.tcn.network[lvl] for e in encs]
        d, p = blocks[0].dilation, blocks[0].p
        y = x
        spec = prev = None
        for ci, name in enumerate(('conv1', 'conv2')):
            convs = [getattr(b, name) for b in blocks]
            ws = wn[lvl][ci]
            defer = ci == 1 or ops.im2col_drop_ok(ws[0].shape[0], wn[lvl][1][0].shape[2])
            spec = ops.make_drop(p, e0.training, x.device, deferred=defer)       # see TemporalBlock.forward
            y = ops.grouped_conv1d_tm(y.view(G, B, T, y.shape[2]), ws, [c.bias for c in convs], dil=d, pad_left=d * (ws[0].shape[2] - 1), To=T,
                                      act=ACT_RELU, drop=spec, in_drop=prev if (prev is not None and prev.deferred) else None).view(G * B, T, -1)
            prev = spec
            if spec is None:
                y = ops.dropout(y, p, e0.training)
        x = ops.add_relu(y, x, drop=spec)
    out = ops.grouped_linear(x.view(G, B * T, x.shape[2]), [e.decoder.weight for e in encs], [e.decoder.bias for e in encs])
    return out.view(G, B, T, -1)


# ---------------------------------------------------------------------------------------------------
# bidirectional GRU parameters (torch.nn.GRU names / init)
# ---------------------------------------------------------------------------------------------------

class BiGRU(nn.Module):
    def __init__(self, input_size, hidden_size, num_layers, dropout=0.0):
        super().__init__()
        if not ops.gru_supported(hidden_size):
            raise ValueError('ha2g_amd: GRU hidden size %d has no HIP instantiation (300, 64, 32)' % hidden_size)
        self.input_size, self.hidden_size, self.num_layers, self.dropout = input_size, hidden_size, num_layers, dropout
        self.grad_slice = None        # (first_row, n_rows) hint set by the fused-chain train step
        self._names = []
        k = 1.0 / math.sqrt(hidden_size)
        for l in range(num_layers):
            inp = input_size if l == 0 else 2 * hidden_size
            for suf in ('', '_reverse'):
                for nm, shp in (('weight_ih', (3 * hidden_size, inp)), ('weight_hh', (3 * hidden_size, hidden_size)),
                                ('bias_ih', (3 * hidden_size,)), ('bias_hh', (3 * hidden_size,))):
                    name = '%s_l%d%s' % (nm, l, suf)
                    setattr(self, name, nn.Parameter(_uniform(torch.empty(*shp), k)))
                    self._names.append(name)

    def flatten_parameters(self):
        pass

    def forward(self, x, hidden=None):
        assert hidden is None, 'h0 is always zero on this path'
        masks = None
        if self.training and self.dropout > 0 and self.num_layers > 1:
            if ops.GRU_MASK_SPEC and x.is_cuda:
                masks = [ops.gru_drop_spec(self.dropout, x.device) for _ in range(self.num_layers - 1)]
            else:
                masks = [ops.dropout_mask((x.shape[0], x.shape[1], 2 * self.hidden_size), self.dropout, x.device)
                         for _ in range(self.num_layers - 1)]
        y = ops.bigru(x, [getattr(self, n) for n in self._names], self.hidden_size, masks, self.grad_slice)
        return y, None


# ---------------------------------------------------------------------------------------------------
# generator (reference model/hierarchy_net.py:55-150)
# ---------------------------------------------------------------------------------------------------

def default_eps(shape, device):
    return torch.randn(shape, device=device)


SPEAKER_ROW_SPLIT = False     # True: the speaker sub-network evaluated separately on the gradient-carrying rows and (no_grad) on the rest (round 2-5)


class Hierarchical_PoseGenerator(nn.Module):
    def __init__(self, args, pose_dim, n_words, word_embed_size, word_embeddings, z_obj=None):
        super().__init__()
        self.pre_length = args.n_pre_poses
        self.gen_length = args.n_poses - args.n_pre_poses
        self.z_obj = z_obj
        self.input_context = args.input_context
        if self.input_context == 'none':
            self.in_size = pose_dim + 1
        elif self.input_context in ('audio', 'text'):
            self.in_size = 32 + pose_dim + 1
        else:
            self.in_size = 32 + 32 + pose_dim + 1
        self.text_encoder = TextEncoderTCN(args, n_words, word_embed_size, pre_trained_embedding=word_embeddings,
                                           dropout=args.dropout_prob)
        self.speaker_embedding = None
        if self.z_obj:
            self.z_size = 16
            self.in_size += self.z_size
            if hasattr(z_obj, 'n_words'):                       # vocab.Vocab of speakers
                self.speaker_embedding = nn.Sequential(Embedding(z_obj.n_words, self.z_size), Linear(self.z_size, self.z_size))
                self.speaker_mu = Linear(self.z_size, self.z_size)
                self.speaker_logvar = Linear(self.z_size, self.z_size)
        self.hidden_size = args.hidden_size
        self.gru = BiGRU(self.in_size, self.hidden_size, args.n_layers, dropout=args.dropout_prob)
        self.out = nn.Sequential(Linear(self.hidden_size, self.hidden_size // 2), _Marker(), Linear(self.hidden_size // 2, pose_dim))
        self.do_flatten_parameters = False
        self.eps_source = default_eps                            # injectable reparameterisation noise (parity tests)

    def _row_split(self, fn, *tensors):
        """Row-wise sub-networks (no BatchNorm in the generator => rows are independent) are evaluated separately on the
        rows that will receive gradient and, under no_grad, on the rest, so their backward touches only the former.
        Active when the fused-chain train step set `gru.grad_slice`; otherwise a plain call."""
        gs = self.gru.grad_slice
        if gs is None or not torch.is_grad_enabled():
            return fn(*tensors)
        s0, cnt = gs
        B = tensors[0].shape[0]
        parts = []
        for a, b, grad in ((0, s0, False), (s0, s0 + cnt, True), (s0 + cnt, B, False)):
            if b <= a:
                continue
            sl = [t[a:b] for t in tensors]
            if grad:
                parts.append(fn(*sl))
            else:
                with torch.no_grad():
                    parts.append(fn(*sl))
        if isinstance(parts[0], tuple):
            return tuple(torch.cat([p[i] for p in parts]) for i in range(len(parts[0])))
        return torch.cat(parts)

    def _speaker(self, vid_indices, eps):
        z_context = self.speaker_embedding(vid_indices)
        z_mu = self.speaker_mu(z_context)
        z_logvar = self.speaker_logvar(z_context)
        return ops.reparameterize(z_mu, z_logvar, eps), z_mu, z_logvar

    def _head(self, gru_out):
        output = ops.dirsum(gru_out)                               # sum bidirectional outputs
        h = self.out[0](output.reshape(-1, output.shape[2]), act=ACT_LEAKY)
        return self.out[2](h).reshape(gru_out.shape[0], gru_out.shape[1], -1)

    def forward(self, pre_seq, in_text, audio_feat_seq=None, vid_indices=None, text_feat_seq=None):
        # text_feat_seq: this generator's text features when the caller evaluated all generators' text encoders together
        # (grouped_text_encoders); None = run the own encoder here, as the reference does
        if self.input_context == 'none':
            text_feat_seq = None
        else:
            if text_feat_seq is None:
                text_feat_seq = self._row_split(self.text_encoder, in_text)
            assert audio_feat_seq.shape[1] == text_feat_seq.shape[1]
        if self.z_obj:
            if self.speaker_embedding:
                assert vid_indices is not None
                eps = self.eps_source((vid_indices.shape[0], self.z_size), vid_indices.device)
                # the speaker MLP (an embedding row + three 16 x 16 Linears per sample) runs ONCE over all fused rows: its backward costs the same launches
                # on 3B rows as on B (the extra rows receive exact zeros and sit in front of the gradient-carrying block: the same partial sums), and the
                # row split's second pass + three torch.cat are saved (round 6: -8 launches per generator)
                z_context, z_mu, z_logvar = (self._row_split if SPEAKER_ROW_SPLIT else (lambda f, *t: f(*t)))(self._speaker, vid_indices, eps)
            else:
                z_mu = z_logvar = None
                z_context = torch.randn(audio_feat_seq.shape[0], self.z_size, device=audio_feat_seq.device)
        else:
            z_mu = z_logvar = z_context = None
        if self.input_context == 'both' and z_context is not None and pre_seq.is_cuda:
            # the configuration hierarchy.yml runs: [pre_seq | audio | text | z over time] in one pack kernel
            output, _ = self.gru(ops.gen_concat(pre_seq, audio_feat_seq, text_feat_seq, z_context), None)
            return self._row_split(self._head, output), z_context, z_mu, z_logvar
        if self.input_context == 'both':
            parts = [pre_seq, audio_feat_seq, text_feat_seq]
        elif self.input_context == 'audio':
            parts = [pre_seq, audio_feat_seq]
        elif self.input_context == 'text':
            parts = [pre_seq, text_feat_seq]
        elif self.input_context == 'none':
            parts = [pre_seq]
        else:
            assert False
        if z_context is not None:
            parts.append(z_context.unsqueeze(1).expand(-1, pre_seq.shape[1], -1))
        in_data = torch.cat(parts, dim=2)                          # layout plumbing (cat / expand) stays in torch
        output, _ = self.gru(in_data, None)
        decoder_outputs = self._row_split(self._head, output)
        return decoder_outputs, z_context, z_mu, z_logvar


# ---------------------------------------------------------------------------------------------------
# discriminator (reference model/hierarchy_net.py:197-242)
# ---------------------------------------------------------------------------------------------------

class _Conv1dParams(nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        b = 1.0 / math.sqrt(cin * k)
        self.weight = nn.Parameter(_uniform(torch.empty(cout, cin, k), b))
        self.bias = nn.Parameter(_uniform(torch.empty(cout), b))


class Hierarchical_ConvDiscriminator(nn.Module):
    def __init__(self, input_size, n_frames=34):
        """n_frames: the reference hard-codes the 34-frame window (out2 = Linear(28, 1), model/hierarchy_net.py:216); other window lengths (SURVEY M5:
        T = 62 is the legal neighbour of BASELINE config 5's T = 64) size out2 as Linear(n_frames - 6, 1) -- performance runs only, no reference twin."""
        super().__init__()
        self.input_size = input_size
        self.hidden_size = 64
        self.pre_conv = nn.Sequential(_Conv1dParams(input_size, 16, 3), BatchNormParams(16), _Marker(),
                                      _Conv1dParams(16, 8, 3), BatchNormParams(8), _Marker(), _Conv1dParams(8, 8, 3))
        self.gru = BiGRU(8, self.hidden_size, 4, dropout=0.3)
        self.out = Linear(self.hidden_size, 1)
        self.out2 = Linear(n_frames - 6, 1)
        self.do_flatten_parameters = False

    def _features(self, poses):
        """pre_conv stack: Conv1d -> BatchNorm1d (train: statistics of THIS call's batch) -> LeakyReLU, twice, then Conv1d"""
        pc = self.pre_conv
        x = ops.conv1d_tm(poses, pc[0].weight, pc[0].bias)                    # [B, T-2, 16]
        x = pc[1](x, act=ACT_LEAKY)
        x = ops.conv1d_tm(x, pc[3].weight, pc[3].bias)
        x = pc[4](x, act=ACT_LEAKY)
        return ops.conv1d_tm(x, pc[6].weight, pc[6].bias)                     # [B, T-6, 8]

    def _classify(self, feat):
        output, _ = self.gru(feat, None)
        output = ops.dirsum(output)
        batch_size = feat.shape[0]
        output = self.out(output.reshape(-1, output.shape[2]))
        output = output.view(batch_size, -1)
        return self.out2(output, act=ACT_SIGMOID)

    def forward(self, poses, in_text=None):
        return self._classify(self._features(poses))

    def forward_pair(self, real, fake, in_text=None):
        """D(real), D(fake) of the discriminator update (train_hierarchy.py:121-128) in one pass through the row-wise part: the two
        pre_conv stacks run separately and in the reference's order (train-mode BatchNorm normalises each call with its own batch
        statistics and updates the running ones), the latency-bound 4-layer GRU and the two Linear heads -- independent per sample --
        see both batches at once (one set of 4 + 4 layer launches instead of two)."""
        feat = torch.cat([self._features(real), self._features(fake)])
        out = self._classify(feat)
        return out[:real.shape[0]], out[real.shape[0]:]


# ---------------------------------------------------------------------------------------------------
# audio encoder (reference model/hierarchy_net.py:10-19, model/ResNetSE34V2.py, model/ResNetBlocks.py)
# ---------------------------------------------------------------------------------------------------

class _Conv2dParams(nn.Module):
    def __init__(self, cin, cout, k, bias):
        super().__init__()
        w = torch.empty(cout, cin, k, k)
        nn.init.kaiming_normal_(w, mode='fan_out', nonlinearity='relu')           # ResNetSE34V2.py:89-91
        self.weight = nn.Parameter(w.contiguous(memory_format=torch.channels_last))   # physical [Cout][KH][KW][Cin]
        if bias:
            self.bias = nn.Parameter(_uniform(torch.empty(cout), 1.0 / math.sqrt(cin * k * k)))


class SELayer(nn.Module):
    def __init__(self, channel, reduction=8):
        super().__init__()
        self.fc = nn.Sequential(Linear(channel, channel // reduction), _Marker(), Linear(channel // reduction, channel), _Marker())


class SEBasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, reduction=8):
        super().__init__()
        self.conv1 = _Conv2dParams(inplanes, planes, 3, False)
        self.bn1 = BatchNormParams(planes)
        self.conv2 = _Conv2dParams(planes, planes, 3, False)
        self.bn2 = BatchNormParams(planes)
        self.se = SELayer(planes, reduction)
        self.downsample = downsample
        self.stride = stride


class ResNetSE(nn.Module):
    def __init__(self, args, block, layers, num_filters, nOut, z_obj, pose_level=3, n_mels=128, **kwargs):
        super().__init__()
        assert tuple(layers) == wav_engine.LAYERS and tuple(num_filters) == wav_engine.FILTERS
        self.pose_level = pose_level
        self.z_obj = z_obj
        self.inplanes = num_filters[0]
        self.conv1 = _Conv2dParams(1, num_filters[0], 3, True)
        self.bn1 = BatchNormParams(num_filters[0])
        self.conv_low = _Conv2dParams(64, 64, 2, True)
        self.bn_low = BatchNormParams(64)
        self.fc_low = Linear(63 * 64, nOut)
        self.conv_mid = _Conv2dParams(32, 32, 3, True)
        self.bn_mid = BatchNormParams(32)
        self.fc_mid = Linear(62 * 32, nOut)
        self.conv_high = _Conv2dParams(16, 16, 3, True)
        self.bn_high = BatchNormParams(16)
        self.fc_high = Linear(62 * 16, nOut)
        self.layer1 = self._make_layer(block, num_filters[0], layers[0])
        self.layer2 = self._make_layer(block, num_filters[1], layers[1], stride=2)
        self.layer3 = self._make_layer(block, num_filters[2], layers[2], stride=2)
        self.layer4 = self._make_layer(block, num_filters[3], layers[3], stride=2)
        assert z_obj is not None and hasattr(z_obj, 'n_words'), 'hierarchy runs use z_type=speaker'
        self.speaker_embedding = nn.Sequential(Embedding(z_obj.n_words, 16), Linear(16, 16))
        self.fc1 = Linear(16, 32)
        self.fc2 = Linear(32, self.pose_level * 3)
        self._names = wav_engine.param_names(pose_level)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(_Conv2dParams(self.inplanes, planes * block.expansion, 1, False),
                                       BatchNormParams(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def _get(self, dotted):
        obj = self
        for part in dotted.split('.'):
            obj = obj[int(part)] if part.isdigit() else getattr(obj, part)
        return obj

    def forward(self, x, vid_indices):
        """x: [B, 1, 128, W] (as the reference passes it) or [B, 128, W]."""
        if x.dim() == 4:
            x = x[:, 0]
        tensors, bufs = [], {}
        objs = self.__dict__.get('_objs')                # the dotted names resolved once (0.8 ms of host time per step otherwise); Parameters keep their identity
        if objs is None:
            objs = [self._get(n) for n in self._names]
            self.__dict__['_objs'] = objs
        for n, obj in zip(self._names, objs):
            if isinstance(obj, BatchNormParams):
                tensors += [obj.weight, obj.bias]
                bufs[n] = (obj.running_mean, obj.running_var, obj.num_batches_tracked)
            else:
                tensors.append(obj)
        outs = wav_engine.WavEncoderFunction.apply(x, vid_indices, self.pose_level, self._names, (bufs, self.training), *tensors)
        weight, low, mid, high = outs[:4]
        return weight, low, mid, high, list(outs[4:])


class Hierarchical_WavEncoder(nn.Module):
    def __init__(self, args, z_obj, pose_level, nOut=32):
        super().__init__()
        self.feat_extractor = ResNetSE(args, SEBasicBlock, [3, 4, 6, 3], [32, 64, 128, 256], nOut=nOut, pose_level=pose_level,
                                       z_obj=z_obj, n_mels=128)

    def forward(self, audio_spectrum, vid_indices):
        return self.feat_extractor(audio_spectrum.unsqueeze(1), vid_indices)

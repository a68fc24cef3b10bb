"""CPU ORACLE of the FGD evaluator -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (only tests/ import it).

Functional restatement of scripts/model/embedding_space_evaluator.py:57-154 and of the eval-mode pose auto-encoder it wraps
(model/embedding_net.py:42-82 PoseEncoderConv, :156-218 PoseDecoderConv, EmbeddingNet 'pose' mode) over a plain state dict.
Parity status: PINNED -- tests/test_fgd.py checks it in float64 against tests/golden/fgd.npz, produced by running the reference's
own EmbeddingSpaceEvaluator (tests/golden/gen_golden.py::fgd_goldens)."""
import numpy as np
import torch
import torch.nn.functional as F
from scipy import linalg


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + 'running_mean'], sd[p + 'running_var'], sd[p + 'weight'], sd[p + 'bias'], False, 0.1, 1e-5)


def encode(poses, sd, enc='pose_encoder.', mu='pose_encoder.fc_mu.'):
    x = poses.transpose(1, 2)
    for i, stride in ((0, 1), (1, 1), (2, 2)):
        x = F.conv1d(x, sd['%snet.%d.0.weight' % (enc, i)], sd['%snet.%d.0.bias' % (enc, i)], stride=stride)
        x = F.leaky_relu(_bn(x, sd, '%snet.%d.1.' % (enc, i)), 0.2)
    x = F.conv1d(x, sd[enc + 'net.3.weight'], sd[enc + 'net.3.bias']).flatten(1)
    x = _bn(F.linear(x, sd[enc + 'out_net.0.weight'], sd[enc + 'out_net.0.bias']), sd, enc + 'out_net.1.')     # LeakyReLU(True) = identity
    x = _bn(F.linear(x, sd[enc + 'out_net.3.weight'], sd[enc + 'out_net.3.bias']), sd, enc + 'out_net.4.')
    x = F.linear(x, sd[enc + 'out_net.6.weight'], sd[enc + 'out_net.6.bias'])
    return F.linear(x, sd[mu + 'weight'], sd[mu + 'bias']) if mu else x


def decode(z, sd, dec='decoder.'):
    x = _bn(F.linear(z, sd[dec + 'pre_net.0.weight'], sd[dec + 'pre_net.0.bias']), sd, dec + 'pre_net.1.')
    x = F.linear(x, sd[dec + 'pre_net.3.weight'], sd[dec + 'pre_net.3.bias']).view(z.shape[0], 4, -1)
    x = F.leaky_relu(_bn(F.conv_transpose1d(x, sd[dec + 'net.0.weight'], sd[dec + 'net.0.bias']), sd, dec + 'net.1.'), 0.2)
    x = F.leaky_relu(_bn(F.conv_transpose1d(x, sd[dec + 'net.3.weight'], sd[dec + 'net.3.bias']), sd, dec + 'net.4.'), 0.2)
    x = F.conv1d(x, sd[dec + 'net.6.weight'], sd[dec + 'net.6.bias'])
    return F.conv1d(x, sd[dec + 'net.7.weight'], sd[dec + 'net.7.bias']).transpose(1, 2)


def recon_metrics(recon, poses):
    rl = torch.mean(torch.abs(recon - poses), dim=(1, 2)) + torch.mean(torch.abs((recon[:, 1:] - recon[:, :-1]) - (poses[:, 1:] - poses[:, :-1])), dim=(1, 2))
    B, T, _ = poses.shape
    cos = torch.sum(1 - torch.cosine_similarity(recon.reshape(B, T, -1, 3), poses.reshape(B, T, -1, 3), dim=-1))
    return torch.sum(rl), cos


class Evaluator:
    def __init__(self, sd, enc='pose_encoder.', dec='decoder.', mu='pose_encoder.fc_mu.'):
        """defaults = EmbeddingNet 'pose' mode (TED-Gesture); MotionAE (TED-Expressive, model/motion_ae.py): enc='encoder.', mu=None"""
        self.sd, self.names = sd, (enc, dec, mu)
        self.real, self.gen, self.recon_err_diff, self.cos_err_diff = [], [], [], []

    def push_samples(self, generated, real):
        enc, dec, mu = self.names
        with torch.no_grad():
            rf, gf = encode(real, self.sd, enc, mu), encode(generated, self.sd, enc, mu)
            rr, gr = decode(rf, self.sd, dec), decode(gf, self.sd, dec)
            a, b = recon_metrics(rr, real), recon_metrics(gr, generated)
        self.real.append(rf.numpy()); self.gen.append(gf.numpy())
        self.recon_err_diff.append(float(b[0] - a[0])); self.cos_err_diff.append(float(b[1] - a[1]))

    def get_scores(self):
        g, r = np.vstack(self.gen), np.vstack(self.real)
        mu1, s1, mu2, s2 = np.mean(g, 0), np.cov(g, rowvar=False), np.mean(r, 0), np.cov(r, rowvar=False)
        covmean, _ = linalg.sqrtm(s1.dot(s2), disp=False)
        covmean = covmean.real if np.iscomplexobj(covmean) else covmean
        d = mu1 - mu2
        return d.dot(d) + np.trace(s1) + np.trace(s2) - 2 * np.trace(covmean), float(np.mean(np.sum(np.abs(r - g), axis=-1)))

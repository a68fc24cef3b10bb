"""FGD evaluator on the device -- MI355X-native mirror of scripts/model/embedding_space_evaluator.py:57-154 with the pose
auto-encoders it wraps (TED-Gesture: model/embedding_net.py:42-82,177-218 EmbeddingNet in 'pose' mode; TED-Expressive: model/motion_ae.py).

Same interface (reset / push_samples / get_no_of_samples / get_scores / get_diversity_scores / calculate_frechet_distance), but
nothing leaves the GPU per batch: the eval-mode encoder / decoder run on the HIP GEMM / conv1d / BatchNorm kernels, and the latent
features' sum and outer-product, the feature L1 distance and the reconstruction diagnostics are accumulated in float64 on the device.
Only get_scores() copies 2 x (D + D*D) doubles back and, like the reference, leaves the 32x32 (128x128) matrix square root to SciPy.
"""
import numpy as np
import torch

from . import ops
from ._lib import check, lib
from .ops import ACT_NONE


def _st():
    return torch.cuda.current_stream().cuda_stream


class _PoseAE:
    """Eval-mode forward of PoseEncoderConv + PoseDecoderConv (34 frames) from a reference state_dict.  Activations are time-major
    [B, T, C]; weights whose input/output index the reference's channel-major flatten / view are permuted once at load time."""

    def __init__(self, sd, enc, dec, mu_key, device):
        f = lambda k: sd[k].detach().to(device=device, dtype=torch.float32).contiguous()
        self.enc_convs = [(f('%snet.%d.0.weight' % (enc, i)), f('%snet.%d.0.bias' % (enc, i)), self._bn(sd, '%snet.%d.1.' % (enc, i), device)) for i in range(3)]
        self.enc_last = (f(enc + 'net.3.weight'), f(enc + 'net.3.bias'))
        w0 = f(enc + 'out_net.0.weight')                               # [256, 384], column index c*12 + t  ->  t*32 + c
        self.enc_fc = [(w0.view(-1, 32, 12).permute(0, 2, 1).reshape(w0.shape[0], 384).contiguous(), f(enc + 'out_net.0.bias'), self._bn(sd, enc + 'out_net.1.', device)),
                       (f(enc + 'out_net.3.weight'), f(enc + 'out_net.3.bias'), self._bn(sd, enc + 'out_net.4.', device)),
                       (f(enc + 'out_net.6.weight'), f(enc + 'out_net.6.bias'), None)]
        self.fc_mu = (f(mu_key + 'weight'), f(mu_key + 'bias')) if mu_key else None
        self.dec_pre = [(f(dec + 'pre_net.0.weight'), f(dec + 'pre_net.0.bias'), self._bn(sd, dec + 'pre_net.1.', device))]
        w3, b3 = f(dec + 'pre_net.3.weight'), f(dec + 'pre_net.3.bias')   # [136, 64], row index c*34 + t  ->  t*4 + c
        self.dec_pre.append((w3.view(4, 34, -1).permute(1, 0, 2).reshape(136, -1).contiguous(), b3.view(4, 34).t().reshape(136).contiguous(), None))
        # ConvTranspose1d(stride 1, no padding) == Conv1d over a fully padded input with the kernel flipped and the channel axes swapped
        ct = lambda k: f(k).permute(1, 0, 2).flip(2).contiguous()
        self.dec_convs = [(ct(dec + 'net.0.weight'), f(dec + 'net.0.bias'), self._bn(sd, dec + 'net.1.', device), 2),
                          (ct(dec + 'net.3.weight'), f(dec + 'net.3.bias'), self._bn(sd, dec + 'net.4.', device), 2),
                          (f(dec + 'net.6.weight'), f(dec + 'net.6.bias'), None, 0), (f(dec + 'net.7.weight'), f(dec + 'net.7.bias'), None, 0)]

    @staticmethod
    def _bn(sd, p, device):
        return tuple(sd[p + k].detach().to(device=device, dtype=torch.float32).contiguous() for k in ('weight', 'bias', 'running_mean', 'running_var'))

    @staticmethod
    def _bn_eval(x, bn, slope=None):
        y = ops.batch_norm_eval(x, bn[0], bn[1], bn[2], bn[3], 1e-5, ACT_NONE)
        return ops.eltwise(ops.OP_LEAKY_A, y, alpha=slope) if slope is not None else y

    def encode(self, poses):
        x = poses.contiguous().float()
        B = x.shape[0]
        for i, (w, b, bn) in enumerate(self.enc_convs):
            if i < 2:
                x = ops.conv1d_tm(x, w, b)
            else:                                                       # kernel 4, stride 2: every second frame of the stride-1 result
                x = ops.conv1d_tm(x, w, b)[:, ::2].contiguous()
            x = self._bn_eval(x, bn, 0.2)
        x = ops.conv1d_tm(x, *self.enc_last)                           # [B, 12, 32]
        x = x.reshape(B, 384)
        for w, b, bn in self.enc_fc:                                   # nn.LeakyReLU(True): negative_slope = 1.0, the identity
            x = ops.linear(x, w, b)
            if bn is not None:
                x = self._bn_eval(x, bn)
        if self.fc_mu is not None:
            x = ops.linear(x, *self.fc_mu)                             # variational_encoding=False: z = mu
        return x

    def decode(self, z):
        x = z
        for w, b, bn in self.dec_pre:
            x = ops.linear(x, w, b)
            if bn is not None:
                x = self._bn_eval(x, bn)
        x = x.view(z.shape[0], 34, 4)
        for w, b, bn, pad in self.dec_convs:
            x = ops.conv1d_tm(x, w, b, pad_left=pad, To=x.shape[1] + 2 * pad - 2)
            if bn is not None:
                x = self._bn_eval(x, bn, 0.2)
        return x                                                       # [B, 34, P]


class EmbeddingSpaceEvaluator:
    def __init__(self, args, embed_net_path, lang_model, device):
        """embed_net_path: the reference's checkpoint file ({'pose_dim', 'gen_dict'} / {'pose_dim', 'latent_dim', 'motion_ae'}) or such a dict."""
        self.n_pre_poses = args.n_pre_poses
        ckpt = embed_net_path if isinstance(embed_net_path, dict) else torch.load(embed_net_path, map_location='cpu')
        self.pose_dim = ckpt['pose_dim']
        self.device = torch.device(device)
        assert args.n_poses == 34, 'the evaluation auto-encoders exist for 34-frame windows'
        if args.pose_dim == 27:
            self.net = _PoseAE(ckpt['gen_dict'], 'pose_encoder.', 'decoder.', 'pose_encoder.fc_mu.', self.device)
            self.feat_dim = 32
        elif args.pose_dim == 126:
            self.latent_dim = ckpt['latent_dim']
            self.net = _PoseAE(ckpt['motion_ae'], 'encoder.', 'decoder.', None, self.device)
            self.feat_dim = self.latent_dim
        else:
            raise ValueError('pose_dim %r' % args.pose_dim)
        self.reset()

    def reset(self):
        D = self.feat_dim
        z = lambda *s: torch.zeros(*s, dtype=torch.float64, device=self.device)
        self._stats = {k: (z(D), z(D, D)) for k in ('real', 'gen')}
        self._l1 = z(1)
        self._n = 0
        self.real_feat_list, self.generated_feat_list, self.context_feat_list = [], [], []      # device tensors, one per batch
        self.recon_err_diff, self.cos_err_diff = [], []

    def get_no_of_samples(self):
        return len(self.real_feat_list)

    def _recon_metrics(self, recon, poses):
        B, T, P = poses.shape
        out = torch.zeros(2 + 2 * B, dtype=torch.float64, device=self.device)
        check(lib.ha2g_recon_metrics_f64(recon.contiguous().data_ptr(), poses.contiguous().data_ptr(), B, T, P, out.data_ptr(), _st()))
        return out[0], out[1]

    @torch.no_grad()
    def push_samples(self, context_text, context_spec, generated_poses, real_poses):
        real_poses = real_poses.to(self.device).float().contiguous()
        generated_poses = generated_poses.to(self.device).float().contiguous()
        real_feat, gen_feat = self.net.encode(real_poses), self.net.encode(generated_poses)
        real_recon, gen_recon = self.net.decode(real_feat), self.net.decode(gen_feat)
        self.real_feat_list.append(real_feat)
        self.generated_feat_list.append(gen_feat)
        for key, f in (('real', real_feat), ('gen', gen_feat)):
            s, o = self._stats[key]
            check(lib.ha2g_feat_stats_f64(f.data_ptr(), f.shape[0], f.shape[1], s.data_ptr(), o.data_ptr(), _st()))
        check(lib.ha2g_l1_rows_f64(real_feat.data_ptr(), gen_feat.data_ptr(), real_feat.numel(), self._l1.data_ptr(), _st()))
        self._n += real_feat.shape[0]
        rl_real, cos_real = self._recon_metrics(real_recon, real_poses)
        rl_fake, cos_fake = self._recon_metrics(gen_recon, generated_poses)
        self.recon_err_diff.append(rl_fake - rl_real)
        self.cos_err_diff.append(cos_fake - cos_real)

    def get_diversity_scores(self):
        feat1 = torch.cat(self.generated_feat_list[:500])
        random_idx = torch.randperm(len(self.generated_feat_list))[:500]
        feat2 = torch.cat([self.generated_feat_list[int(x)] for x in random_idx])
        if feat1.shape != feat2.shape:                   # > 500 pushed batches with a ragged last one: the reference's numpy expression raises here too
            raise ValueError('get_diversity_scores: operands could not be broadcast together with shapes %s %s' % (tuple(feat1.shape), tuple(feat2.shape)))
        acc = torch.zeros(1, dtype=torch.float64, device=self.device)
        check(lib.ha2g_l1_rows_f64(feat1.data_ptr(), feat2.data_ptr(), feat1.numel(), acc.data_ptr(), _st()))
        return float(acc.item()) / feat1.shape[0]

    def _moments(self, key):
        s, o = (t.cpu().numpy() for t in self._stats[key])
        n = self._n
        mu = s / n
        return mu, (o - n * np.outer(mu, mu)) / (n - 1)               # == np.mean(axis=0), np.cov(rowvar=False) in float64

    def get_scores(self):
        g_mu, g_sigma = self._moments('gen')
        r_mu, r_sigma = self._moments('real')
        try:
            frechet_dist = self.calculate_frechet_distance(g_mu, g_sigma, r_mu, r_sigma)
        except ValueError:
            frechet_dist = 1e+10
        return frechet_dist, float(self._l1.item()) / self._n

    @staticmethod
    def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
        """d^2 = |mu1 - mu2|^2 + Tr(C1 + C2 - 2 sqrt(C1 C2)) -- host side, as in the reference (:156-209; scipy.linalg.sqrtm)."""
        from scipy import linalg
        mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
        sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
        assert mu1.shape == mu2.shape and sigma1.shape == sigma2.shape
        diff = mu1 - mu2
        covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
        if not np.isfinite(covmean).all():
            offset = np.eye(sigma1.shape[0]) * eps
            covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
        if np.iscomplexobj(covmean):
            if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
                raise ValueError('Imaginary component {}'.format(np.max(np.abs(covmean.imag))))
            covmean = covmean.real
        return diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean)

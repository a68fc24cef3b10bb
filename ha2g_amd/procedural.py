"""Closed-form ("procedural") parameters and seeded synthetic batches.

Parity fixtures must stay small, so neither the 5-13 M parameter tensors nor the
inputs are committed: both sides of every parity test (the golden generator that
imports the reference in the build container, and the GPU-side tests that never
see the reference) regenerate them from this module.  Only NumPy's PCG64 stream
is used, which is bit-reproducible across platforms.

Input shapes follow the loader contract of the reference train loop
(scripts/train.py:256-270): in_text_padded int64 (B,T), in_spec f32 (B,128,W),
target f32 (B,T,P), vid_indices int64 (B,).
"""
import zlib

import numpy as np

_TCN_ALIASES = (('.net.0.', '.conv1.'), ('.net.4.', '.conv2.'))


def canonical_key(key):
    """TemporalBlock registers each conv twice (tcn.py:19,25,31): hash the conv1/conv2 name."""
    for a, b in _TCN_ALIASES:
        key = key.replace(a, b)
    return key


def _rng(key, seed):
    return np.random.Generator(np.random.PCG64([zlib.crc32(canonical_key(key).encode()), seed]))


def tensor_for(key, shape, seed=0):
    """Deterministic float32 values for the state_dict entry `key` of `shape`."""
    shape = tuple(int(s) for s in shape)
    r = _rng(key, seed)
    leaf = key.rsplit('.', 1)[-1]
    if leaf == 'num_batches_tracked':
        return np.zeros(shape, np.int64)
    if leaf == 'running_mean':
        return (0.1 * r.standard_normal(shape)).astype(np.float32)
    if leaf == 'running_var':
        return (1.0 + 0.2 * r.random(shape)).astype(np.float32)
    if leaf == 'weight_g':      # weight-norm gain, positive
        return (0.5 + r.random(shape)).astype(np.float32)
    if 'embedding' in key and len(shape) == 2 and leaf == 'weight' and shape[1] in (16, 300):
        return (0.5 * r.standard_normal(shape)).astype(np.float32)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return ((2.0 * r.random(shape) - 1.0) * np.sqrt(3.0 / fan_in)).astype(np.float32)
    if leaf == 'weight':        # 1-D "weight" = BatchNorm gamma
        return (1.0 + 0.2 * (2.0 * r.random(shape) - 1.0)).astype(np.float32)
    return (0.1 * (2.0 * r.random(shape) - 1.0)).astype(np.float32)   # biases, BN beta


def block_input(key, shape, seed=0):
    """A post-ReLU-like feature map (what an SEBasicBlock / tap of the audio tower sees): relu(N(0.3, 1)) in NCHW order."""
    r = _rng(key, seed)
    return np.maximum(r.standard_normal(tuple(shape)) + 0.3, 0.0).astype(np.float32)


def fill_module(module, seed=0, prefix=''):
    """Overwrite every parameter/buffer of a torch module in place (same keys on both sides)."""
    import torch
    sd = module.state_dict()
    new = {k: torch.from_numpy(tensor_for(prefix + k, v.shape, seed)).to(v.dtype) for k, v in sd.items()}
    module.load_state_dict(new)
    return module


def make_batch(B, P, n_words, n_spk, seed, T=34, W=70):
    """Seeded synthetic batch: spec ~ U(-80,0) dB, target ~ N(0,0.1^2), sparse word ids."""
    r = np.random.Generator(np.random.PCG64([seed, 77]))
    spec = (-80.0 * r.random((B, 128, W))).astype(np.float32)
    target = (0.1 * r.standard_normal((B, T, P))).astype(np.float32)
    text = np.zeros((B, T), np.int64)
    lo = min(4, n_words - 1)
    for b in range(B):
        n = int(r.integers(5, 9))
        pos = r.choice(T, size=n, replace=False)
        text[b, pos] = r.integers(lo, n_words, size=n)
    vid = r.integers(1, n_spk, size=(B,)).astype(np.int64)
    return text, spec, target, vid


class EpsStream:
    """Replays reparameterisation noise: call k returns the k-th seeded N(0,1) tensor."""

    def __init__(self, seed):
        self.seed = seed
        self.k = 0

    def __call__(self, shape):
        r = np.random.Generator(np.random.PCG64([self.seed, 991, self.k]))
        self.k += 1
        return r.standard_normal(tuple(shape)).astype(np.float32)


def fixed_perm(n, seed):
    return np.random.Generator(np.random.PCG64([seed, 313])).permutation(n).astype(np.int64)


def sample_of(t, n=64):
    """(l2 norm, strided sample) digest of a large array, used for grads/params in fixtures."""
    a = np.asarray(t, dtype=np.float64).reshape(-1)
    stride = max(1, a.size // n)
    return np.float64(np.sqrt((a * a).sum())), a[::stride][:n].astype(np.float32)


def synth_spectrogram(n_audio, seed):
    """(128, 1 + n_audio // 512) log-mel-like array in [-80, 0] dB for the synthesis fixture."""
    r = np.random.Generator(np.random.PCG64([seed, 55]))
    return (-80.0 * r.random((128, 1 + n_audio // 512))).astype(np.float32)


def synth_words(clip_seconds, n, seed):
    """[(text 'w<k>', start_s, end_s)] sorted by onset: the word list a transcript aligner would hand to the synthesis loop."""
    r = np.random.Generator(np.random.PCG64([seed, 56]))
    starts = np.sort(r.random(n) * (clip_seconds - 0.5))
    return [('w%d' % int(r.integers(0, 30)), float(s), float(s + 0.1 + 0.3 * r.random())) for s in starts]


def fgd_batch(B, i, seed, P=27):
    """(real, generated) pose windows [B, 34, P] of batch i for the FGD fixture: smooth random walks; `generated` = a noisier, biased copy."""
    r = np.random.Generator(np.random.PCG64([seed, 61, i] if P == 27 else [seed, 61, i, P]))
    real = np.cumsum(0.05 * r.standard_normal((B, 34, P)), axis=1) + 0.3 * r.standard_normal((B, 1, P))
    gen = 0.8 * real + 0.1 * r.standard_normal((B, 34, P)) + 0.05
    return real.astype(np.float32), gen.astype(np.float32)


def eval_speakers(n, n_spk, seed):
    """Deterministic stand-in for the validation loop's random speaker draws (ids 1 .. n_spk-1)."""
    r = np.random.Generator(np.random.PCG64([seed, 81]))
    return [int(v) for v in r.integers(1, n_spk, size=n)]

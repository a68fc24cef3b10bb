"""CPU restatement (NumPy, float64) of the log-mel front-end the reference computes OFFLINE with librosa:

    scripts/utils/data_utils.py:34-38            extract_melspectrogram(y, sr=16000)
    dataset_script/script/make_ted_dataset.py:121-123   (same two calls, per clip)
        melspec = librosa.feature.melspectrogram(y=y, sr=16000, n_fft=1024, hop_length=512, power=2)
        log_melspec = librosa.power_to_db(melspec, ref=np.max).astype('float16')          # mels x time

TEST INFRASTRUCTURE ONLY (imported by tests/ and nothing else).

*** PARITY UNPINNED ***  librosa is a third-party dependency (requirements.txt: `librosa`, no version pin) that is not
vendored in the reference and not installed in this image, and the reference holds no golden spectrogram.  What follows
restates librosa's PUBLISHED defaults for that call (0.8/0.9 line, current when the reference was written):
  stft: n_fft = win_length = 1024, hop 512, window = periodic Hann (scipy.signal.get_window('hann', 1024, fftbins=True)),
        center=True with pad_mode='reflect' (librosa >= 0.10 changed the default to 'constant' = zeros; `pad_mode` selects),
        frames = 1 + len(y) // 512;
  mel:  n_mels = 128, fmin = 0, fmax = sr/2, htk=False (Slaney scale: linear < 1 kHz with 200/3 Hz per mel, log above with
        step ln(6.4)/27), triangular filters, norm='slaney' (2 / (f[m+2] - f[m]));
  power_to_db: 10 log10(max(S, 1e-10)) - 10 log10(max(1e-10, max(S))), floored at (max - 80 dB).
The STFT half is cross-checked against scipy.signal.stft (an independent implementation) in tests/test_logmel_oracle.py.
"""
import numpy as np

N_FFT, HOP, N_MELS = 1024, 512, 128


def hann_periodic(n=N_FFT):
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr=16000, n_fft=N_FFT, n_mels=N_MELS):
    """[n_mels][1 + n_fft/2] Slaney-normalised triangular filters (librosa.filters.mel defaults)."""
    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(0.0), hz_to_mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w


def stft_power(y, pad_mode='reflect'):
    """|STFT|^2, [513][frames], center=True."""
    y = np.asarray(y, dtype=np.float64)
    yp = np.pad(y, N_FFT // 2, mode=pad_mode)
    T = 1 + len(y) // HOP
    w = hann_periodic()
    frames = np.stack([yp[t * HOP:t * HOP + N_FFT] * w for t in range(T)], 0)
    return (np.abs(np.fft.rfft(frames, axis=1)) ** 2).T


def power_to_db(S, amin=1e-10, top_db=80.0):
    ref = np.max(S)
    log_spec = 10.0 * np.log10(np.maximum(amin, S)) - 10.0 * np.log10(np.maximum(amin, ref))
    return np.maximum(log_spec, log_spec.max() - top_db)


def extract_melspectrogram(y, sr=16000, pad_mode='reflect', f16=True):
    """data_utils.py:34-38: (128, 1 + len(y)//512) log-mel in dB relative to the clip maximum."""
    S = mel_filterbank(sr) @ stft_power(y, pad_mode)
    db = power_to_db(S)
    return db.astype(np.float16) if f16 else db

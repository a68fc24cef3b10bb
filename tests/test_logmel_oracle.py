"""The log-mel oracle (oracle/logmel_oracle.py, parity UNPINNED: librosa is absent) against what CAN be checked here:
its STFT against scipy.signal.stft (independent implementation), the Slaney mel scale against its defining constants, and
the shape / range contract of the reference's loader (scripts/data_loader/lmdb_data_loader.py:70,158: 34 frames at 15 fps
-> 70 spectrogram columns; values in [-80, 0] dB)."""
import numpy as np
import scipy.signal

from oracle import logmel_oracle as L


def _clip(n=36267, seed=3):
    r = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n) / 16000.0
    return 0.3 * np.sin(2 * np.pi * 440 * t) + 0.1 * np.sin(2 * np.pi * 3000 * t + 1.0) + 0.02 * r.standard_normal(n)


def test_stft_matches_scipy():
    y = _clip()
    P = L.stft_power(y, 'reflect')
    f, t, Z = scipy.signal.stft(y, fs=16000, window='hann', nperseg=1024, noverlap=512, boundary='even', padded=False)
    Z = Z * L.hann_periodic().sum()                     # scipy scales by 1 / sum(window)
    n = min(P.shape[1], Z.shape[1])
    assert n >= P.shape[1] - 1
    ref = np.abs(Z[:, :n]) ** 2
    assert np.allclose(P[:, :n], ref, rtol=1e-9, atol=1e-9 * ref.max())


def test_slaney_mel_scale_constants():
    assert abs(float(L.hz_to_mel(1000.0)) - 15.0) < 1e-12
    assert abs(float(L.hz_to_mel(6400.0)) - 42.0) < 1e-9          # one "log step" region: 27 mels per factor 6.4
    assert np.allclose(L.mel_to_hz(L.hz_to_mel([0.0, 250.0, 999.0, 1000.0, 4000.0, 8000.0])), [0.0, 250.0, 999.0, 1000.0, 4000.0, 8000.0])
    fb = L.mel_filterbank()
    assert fb.shape == (128, 513) and (fb >= 0).all()
    # every filter is a single triangle; Slaney norm gives unit area in Hz (up to FFT-bin discretisation)
    df = 8000.0 / 512
    area = fb.sum(1) * df
    assert np.all(np.abs(area - 1.0) < 0.35) and abs(np.median(area) - 1.0) < 0.02
    peaks = fb.argmax(1)
    assert (np.diff(peaks) >= 0).all() and peaks[0] >= 1 and peaks[-1] <= 511


def test_shape_and_range_contract():
    y = _clip()
    m = L.extract_melspectrogram(y)
    assert m.dtype == np.float16 and m.shape == (128, 1 + len(y) // 512)
    assert m.shape[1] >= 70                               # loader crops to calc_spectrogram_length_from_motion_length(34, 15) = 70
    assert float(m.max()) == 0.0 and float(m.min()) >= -80.0
    peak_mel = int(np.unravel_index(np.argmax(m.astype(np.float32)), m.shape)[0])
    fb = L.mel_filterbank()
    assert abs(int(fb[:, round(440 / (8000 / 512))].argmax()) - peak_mel) <= 1     # the 440 Hz tone dominates
    z = L.extract_melspectrogram(y, pad_mode='constant', f16=False)
    assert np.allclose(z[:, 2:-2], L.extract_melspectrogram(y, f16=False)[:, 2:-2], atol=1e-9)   # padding only touches the edge frames

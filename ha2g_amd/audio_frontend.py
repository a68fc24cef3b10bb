"""On-GPU log-mel front-end: the reference computes its (128, T) spectrograms offline with librosa
(scripts/utils/data_utils.py:34-38 `extract_melspectrogram`, dataset_script/script/make_ted_dataset.py:121-123) and the
loader crops them to 70 columns (scripts/data_loader/lmdb_data_loader.py:70,158,170).  This module produces the same
arrays from raw 16 kHz audio on the device, batched, so a data pipeline can ship audio instead of spectrograms.

Parity is UNPINNED: librosa is an un-vendored third-party dependency of the reference that is absent from this image;
the kernels are checked against oracle/logmel_oracle.py (a restatement of librosa's published defaults)."""
import torch

from ._lib import check, lib

_tables = {}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def tables(device, sr=16000):
    key = (device.type, device.index, sr)
    if key not in _tables:
        t = torch.empty(lib.ha2g_logmel_tables_floats(), dtype=torch.float32, device=device)
        check(lib.ha2g_logmel_init_f32(t.data_ptr(), sr, _stream()))
        _tables[key] = t
    return _tables[key]


def batch_log_mel(audio, sr=16000, pad_mode='reflect', f16=True):
    """audio [B, n] float32 on a GPU -> [B, 128, 1 + n // 512] log-mel (dB relative to each clip's maximum, >= -80)."""
    assert audio.is_cuda and audio.dtype == torch.float32 and audio.dim() == 2
    assert pad_mode in ('reflect', 'constant')
    audio = audio.contiguous()
    B, n = audio.shape
    T = lib.ha2g_logmel_frames(n)
    out = torch.empty(B, 128, T, dtype=torch.float32, device=audio.device)
    ws = torch.empty(lib.ha2g_logmel_workspace_floats(B, n), dtype=torch.float32, device=audio.device)
    check(lib.ha2g_logmel_f32(audio.data_ptr(), B, n, int(pad_mode == 'reflect'), tables(audio.device, sr).data_ptr(), int(f16),
                              out.data_ptr(), ws.data_ptr(), _stream()))
    return out


def extract_melspectrogram(y, sr=16000):
    """Drop-in for scripts/utils/data_utils.py:34-38: one clip (1-D tensor / array) -> (128, T) float16 tensor on the device."""
    y = torch.as_tensor(y, dtype=torch.float32)
    if not y.is_cuda:
        y = y.cuda()
    return batch_log_mel(y.reshape(1, -1), sr)[0].to(torch.float16)

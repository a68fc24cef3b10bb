"""Throughput of the on-GPU log-mel front-end at the train batch (B=128 clips of 36 267 samples)."""
import sys, torch
sys.path.insert(0, '.')
from ha2g_amd import audio_frontend as fe
dev = torch.device('cuda:0')
for B in (128, 1024):
    a = torch.randn(B, 36267, device=dev) * 0.1
    for _ in range(3): fe.batch_log_mel(a)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): fe.batch_log_mel(a)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print('B=%d: %.3f ms per batch = %.0f clips/s = %.1f x real time per clip-second... (%.2f GB/s of audio)' % (B, ms, B / ms * 1e3, B * 36267 / 16000 / (ms * 1e-3), B * 36267 * 4 / ms / 1e6))

"""Forward trace of the bf16-storage trunk vs the oracle with the same rounding points: per block output, fraction of bit-equal elements and
rms-relative distance (debug aid for tests/test_gpu_b16.py)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from ha2g_amd import wav_b16 as wb, wav_engine as we
from ha2g_amd.config import CASES
from ha2g_testing import batch_for, engine_P, state_for
from oracle import ha2g_oracle as O

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
case = dict(CASES['small'], B=B)
sd = state_for(case)
_, spec, _, vid = batch_for(case)
# oracle trace
trace = {}
orig = O.se_block


def rec(x, sd_, p, stride, has_down, update_bn=True):
    y = orig(x, sd_, p, stride, has_down, update_bn)
    trace[p.split('feat_extractor.')[1]] = y.detach()
    return y


O.se_block = rec
sd_o = {k: v.clone() for k, v in sd.items()}
with O.bf16_storage(), torch.no_grad():
    O.wav_encoder(spec, vid, sd_o, 'audio.', 3)
O.se_block = orig
# HIP trace
q = 'audio.feat_extractor.'
P = engine_P({k[len(q):]: v for k, v in sd.items() if k.startswith(q)}, 'cuda:0')
with torch.no_grad():
    feats, S = wb.trunk_fwd(spec.to('cuda:0').contiguous(), P, we.LAYERS, True, [])
print('%-12s %10s %10s %10s' % ('block', 'equal', 'rms_rel', 'max_rel'))
c0 = S['stem'][1]
for li, nblk in enumerate(we.LAYERS):
    for j in range(nblk):
        b = 'layer%d.%d.' % (li + 1, j)
        h = S[b][15].float().permute(0, 3, 1, 2).cpu()
        o = trace[b]
        print('%-12s %10.6f %10.2e %10.2e' % (b, float((h == o).float().mean()), float((h - o).norm() / o.norm()), float((h - o).abs().max() / o.abs().max())))

"""Dump the digests (norm + 64 strided samples, as tests/golden/gen_golden.digest) of every audio-encoder gradient of step 0 of the `cfg1` case,
computed by the HIP step in the given matrix-core modes -> gpurun_out/cfg1_audio_mode<m>.npz.  Input of the tail-study analysis
(tests/golden/cfg1_tail.npz): which forward variants are inside the reference's own fp32 scatter.   usage: python tools/dump_cfg1_audio_grads.py 6 0 14"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from ha2g_amd import procedural as proc, train_hierarchy as th
from ha2g_amd._lib import DEFAULT_GEMM_MODE, lib
from ha2g_amd.config import CASES
from ha2g_amd.optim import FusedAdam
from ha2g_testing import EpsInjector, batch_for, build_modules

DEV = 'cuda:0'
case = CASES['cfg1']
for mode in [int(a) for a in sys.argv[1:]] or [6, 0, 14]:
    lib.ha2g_gemm_set_mode(mode)
    args, gens, dis, aud, txt = build_modules(case, DEV)
    text, spec, target, vid = (t.to(DEV) for t in batch_for(case))
    lr = float(args.learning_rate)
    g_opts = [FusedAdam(m.parameters(), lr=lr) for m in gens]
    dis_opt = FusedAdam(dis.parameters(), lr=lr * args.discriminator_lr_weight)
    aud_opt, txt_opt = FusedAdam(aud.parameters(), lr=lr), FusedAdam(txt.parameters(), lr=lr)
    EpsInjector(gens, case['seed'], case['B'])
    perm = torch.from_numpy(proc.fixed_perm(case['B'], case['seed'])).to(DEV)
    th.randperm_source = lambda n, device: perm
    try:
        th.train_iter_hierarchy(args, 0, text, spec, target, vid, *gens, dis, aud, txt, *g_opts, dis_opt, aud_opt, txt_opt)
    finally:
        th.randperm_source = None
        lib.ha2g_gemm_set_mode(DEFAULT_GEMM_MODE)
    out = {}
    for k, p in aud.named_parameters():
        if p.grad is None:
            continue
        a = p.grad.detach().double().cpu().numpy().reshape(-1) if p.grad.dim() != 4 else p.grad.detach().double().cpu().contiguous().numpy().reshape(-1)
        out['step0/grad/audio.%s/norm' % k] = np.float64(np.sqrt((a * a).sum()))
        out['step0/grad/audio.%s/sample' % k] = a[::max(1, a.size // 64)][:64].copy()
    np.savez_compressed('gpurun_out/cfg1_audio_mode%d.npz' % mode, **out)
    print('mode', mode, len(out), 'arrays')

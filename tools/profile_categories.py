#!/usr/bin/env python3
"""Group a tools/rocpd_stats.py summary into kernel families (per-step numbers).  usage: profile_categories.py stats.txt n_steps"""
import re
import sys
rows = []
for l in open(sys.argv[1]).read().splitlines()[2:]:
    m = re.match(r'(\S+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)%', l)
    if m:
        rows.append((m.group(1), int(m.group(2)), float(m.group(3))))
steps = int(sys.argv[2])
def cat(n):
    if 'gemm_kernel' in n:
        mm = re.search(r'ILi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)', n)
        am, bm = int(mm.group(5)), int(mm.group(6))
        return 'conv fwd/dgrad (implicit GEMM)' if am == 2 else ('conv wgrad (implicit GEMM)' if bm == 2 else 'dense GEMM')
    if 'gemm_x3_kernel' in n:
        mm = re.search(r'ILi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)', n)
        return 'conv dgrad (split implicit GEMM, gemm_x3)' if int(mm.group(5)) == 2 else 'dense GEMM'
    for key, name in (('pconv_r_kernel', 'patch-resident plane kernel (3x3 stride-1 conv fwd / dgrad of layers 2-4)'), ('pconv_q_kernel', 'plane kernel (1x1 / stride-2 convs + large dense products, 16x16x32)'), ('pack_whh', 'layout (shuffle/pack/permute)'),
                      ('se_bn_', 'SE + bn2 backward in two passes (round 6)'), ('se_mlp', 'SE pointwise'), ('gen_concat', 'pointwise (act bwd, adds, masks)'), ('step_inc', 'Adam'),
                      ('pconv_kernel', 'conv fwd + dgrad (planes, DMA-staged)'), ('pconv_pp_kernel', 'conv fwd + dgrad (planes, DMA-staged)'), ('pconv_wgrad', 'conv wgrad (planes, DMA-staged)'),
                      ('weight_ihwo_planes', 'layout (shuffle/pack/permute)'), ('f32_to_planes', 'layout (shuffle/pack/permute)'),
                      ('conv3x3_c32_wgrad', 'conv 32ch direct (fwd / dgrad / wgrad)'), ('conv3x3_c32pp', 'conv 32ch direct (fwd / dgrad / wgrad)'), ('conv3x3_x3p', 'conv 32ch direct (fwd / dgrad / wgrad)'), ('conv3x3_c32_kernel', 'conv 32ch direct (fwd / dgrad / wgrad)'),
                      ('conv3x3_x3_kernel', 'conv 32ch direct (fwd / dgrad / wgrad)'), ('pool_final', 'SE pointwise'), ('transpose_batched', 'layout (shuffle/pack/permute)'),
                      ('gru_', 'GRU recurrences'), ('splitk', 'split-K reduce'), ('colsum', 'bias-grad column sums'),
                      ('col_partial', 'BatchNorm'), ('bn_', 'BatchNorm'), ('pair_final', 'BatchNorm'),
                      ('image_col', 'SE pointwise'), ('se_scale', 'SE pointwise'), ('se_bwd', 'SE pointwise'),
                      ('contrastive', 'contrastive loss'), ('rownorm', 'contrastive loss'), ('at6native', 'torch plumbing (add/fill/copy/cat)'),
                      ('rocclr', 'torch plumbing (add/fill/copy/cat)'), ('eltwise', 'pointwise (act bwd, adds, masks)'),
                      ('dropout', 'dropout'), ('im2col', 'im2col/col2im'), ('col2im', 'im2col/col2im'), ('embedding', 'embedding'),
                      ('adam', 'Adam'), ('stem_', 'stem conv'), ('pixel_shuffle', 'layout (shuffle/pack/permute)'),
                      ('nhwc_to', 'layout (shuffle/pack/permute)'), ('ohwi', 'layout (shuffle/pack/permute)'), ('weight_norm', 'weight norm'),
                      ('dirsum', 'pointwise (act bwd, adds, masks)'), ('blend', 'blend'), ('huber', 'small losses'), ('sum_kernel', 'small losses'),
                      ('phys', 'small losses'), ('kld', 'small losses'), ('divreg', 'small losses'), ('gan_', 'small losses')):
        if key in n:
            return name
    return 'other: ' + n[:40]
agg = {}
for n, c, ms in rows:
    a = agg.setdefault(cat(n), [0, 0.0])
    a[0] += c
    a[1] += ms
tot = sum(v[1] for v in agg.values())
print('%-40s %10s %12s %7s' % ('family', 'launches', 'ms/step', 'share'))
for k, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-40s %10d %12.2f %6.1f%%' % (k, c // steps, ms / steps, 100 * ms / tot))
print('%-40s %10d %12.2f' % ('total', sum(v[0] for v in agg.values()) // steps, tot / steps))

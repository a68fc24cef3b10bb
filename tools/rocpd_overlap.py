#!/usr/bin/env python3
"""Queue-overlap trace of ONE kernel inside the step: for every launch of kernels matching `pattern` in the second half of a rocprofv3
kernel trace of bench.py -- its duration, how much of it ran beside kernels of the OTHER queue, and which families those were; then the launches grouped by
their dominant partner family (mean duration per group, against the kernel's duration with nothing beside it).
usage: rocpd_overlap.py results.db total_steps pattern"""
import re
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
steps, pat = int(sys.argv[2]), sys.argv[3]
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in db.execute('pragma table_info(%s)' % kd)]
scol = [r[1] for r in db.execute('pragma table_info(%s)' % ks)]
name_col = 'kernel_name' if 'kernel_name' in scol else 'display_name'
qcol = 'queue_id' if 'queue_id' in cols else 'stream_id'
rows = db.execute('select s.%s, d.start, d.end, d.%s from %s d join %s s on d.kernel_id = s.id order by d.start' % (name_col, qcol, kd, ks)).fetchall()
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[0]]
per = len(adam) // steps
rows = rows[adam[per * (steps // 2) - 1] + 1:]


def fam(n):
    n = re.sub(r'\(.*', '', n)
    for key, name in (('pconv_r_kernel', 'patch-resident plane kernel'), ('pconv_q_kernel', 'q plane kernel'), ('pconv_wgrad', 'plane wgrad'), ('conv3x3_c32_wgrad', '32-channel wgrad'),
                      ('conv3x3_c32pp', '32-channel fwd/dgrad'), ('gru_', 'GRU'), ('splitk', 'split-K reduce'), ('col_partial', 'BatchNorm statistics'), ('bn_bwd_apply', 'BatchNorm bwd apply'),
                      ('bn_', 'BatchNorm other'), ('pair_final', 'BatchNorm other'), ('image_col', 'SE pass'), ('se_', 'SE pass'), ('pool_final', 'SE pass'),
                      ('planes', 'planes split/layout'), ('gemm_kernel', 'dense / implicit GEMM')):
        if key in n:
            return name
    return 'other'


mine = [r for r in rows if pat in r[0]]
if not mine:
    sys.exit('no launch matches %r' % pat)
q_me = mine[0][3]
others = [r for r in rows if r[3] != q_me]
groups, recs = {}, []
j0 = 0
for n, s, e, _ in mine:
    while j0 < len(others) and others[j0][2] <= s:
        j0 += 1
    ov = {}
    j = j0
    while j < len(others) and others[j][1] < e:
        o = min(e, others[j][2]) - max(s, others[j][1])
        if o > 0:
            ov[fam(others[j][0])] = ov.get(fam(others[j][0]), 0) + o
        j += 1
    dur = e - s
    tot = sum(ov.values())
    dom = max(ov, key=ov.get) if tot > 0.25 * dur else 'nothing (< 25 % overlapped)'
    recs.append((dur, tot, dom))
    g = groups.setdefault(dom, [0, 0, 0])
    g[0] += 1; g[1] += dur; g[2] += tot
print('%d launches of *%s* on queue %s in %d steps: mean %.1f us, %.0f %% of their time beside the other queue' % (
    len(mine), pat, q_me, steps - steps // 2, sum(r[0] for r in recs) / len(recs) / 1e3, 100.0 * sum(r[1] for r in recs) / sum(r[0] for r in recs)))
print('%-34s %9s %12s %18s' % ('dominant partner on the other queue', 'launches', 'mean us', 'overlapped share'))
for k, (c, d, o) in sorted(groups.items(), key=lambda kv: -kv[1][0]):
    print('%-34s %9d %12.1f %17.0f%%' % (k, c, d / c / 1e3, 100.0 * o / d))

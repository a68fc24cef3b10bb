#!/usr/bin/env python3
"""Per-queue view of a rocprofv3 kernel trace of bench.py: for the last `steps` train steps, each queue's kernel time by family, its idle
time inside the step span, and how much of it ran beside a kernel of another queue.  usage: rocpd_queues.py results.db total_steps"""
import re
import sqlite3
import subprocess
import sys
db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in db.execute('pragma table_info(%s)' % kd)]
scol = [r[1] for r in db.execute('pragma table_info(%s)' % ks)]
name_col = 'kernel_name' if 'kernel_name' in scol else 'display_name'
qcol = 'queue_id' if 'queue_id' in cols else 'stream_id'
rows = db.execute('select s.%s, d.start, d.end, d.%s from %s d join %s s on d.kernel_id = s.id order by d.start' % (name_col, qcol, kd, ks)).fetchall()
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[0]]
per = len(adam) // steps                      # Adam launches per step
cut = adam[per * (steps // 2) - 1] + 1        # second half of the run
rows = rows[cut:]
nsteps = steps - steps // 2
span = rows[-1][2] - rows[0][1]
sys.path.insert(0, 'tools')


def fam(n):
    n = re.sub(r'\(.*', '', n)
    for key, name in (('pconv_kernel', 'conv fwd + dgrad planes'), ('pconv_pp_kernel', 'conv fwd + dgrad planes'), ('pconv_wgrad', 'conv wgrad planes'), ('gemm_x3', 'conv dgrad x3'), ('conv3x3_c32', 'conv c32 direct'),
                      ('conv3x3_x3', 'conv c32 direct'), ('gru_', 'GRU'), ('splitk', 'split-K reduce'), ('col_partial', 'BatchNorm'), ('bn_', 'BatchNorm'),
                      ('pair_final', 'BatchNorm'), ('image_col', 'SE'), ('se_', 'SE'), ('pool_final', 'SE'), ('planes', 'planes split/layout'),
                      ('adam', 'Adam'), ('eltwise', 'pointwise'), ('dropout', 'pointwise'), ('im2col', 'im2col'), ('col2im', 'im2col')):
        if key in n:
            return name
    if 'gemm_kernel' in n:
        mm = re.search(r'ILi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELb\dELi\d+ELi(\d)', n)
        am, bm, sp = int(mm.group(5)), int(mm.group(6)), int(mm.group(7))
        if am == 2:
            return 'conv fwd (fp32 implicit)' if sp == 0 else 'conv dgrad implicit split'
        if bm == 2:
            return 'conv wgrad implicit'
        return 'dense GEMM fp32' if sp == 0 else 'dense GEMM split (bwd)'
    return 'other'


qs = sorted({r[3] for r in rows}, key=lambda q: -sum(r[2] - r[1] for r in rows if r[3] == q))
print('steps analysed %d, span %.2f ms/step' % (nsteps, span / nsteps / 1e6))
ev = sorted([(r[1], 1, r[3]) for r in rows] + [(r[2], -1, r[3]) for r in rows])
for q in qs[:3]:
    mine = [r for r in rows if r[3] == q]
    busy = sum(r[2] - r[1] for r in mine)
    # overlap with other queues
    depth_other, depth_me, last, both = 0, 0, ev[0][0], 0
    for t, d, qq in ev:
        if depth_me > 0 and depth_other > 0:
            both += t - last
        if qq == q:
            depth_me += d
        else:
            depth_other += d
        last = t
    print('queue %s: %d kernels/step, busy %.2f ms/step (%.0f%% of the step), of which %.2f ms beside another queue\'s kernel' % (
        q, len(mine) // nsteps, busy / nsteps / 1e6, 100.0 * busy / span, both / nsteps / 1e6))
    agg = {}
    for n, s, e, _ in mine:
        a = agg.setdefault(fam(n), [0, 0])
        a[0] += 1; a[1] += e - s
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print('      %-28s %5d  %7.2f ms/step' % (k, c // nsteps, t / nsteps / 1e6))

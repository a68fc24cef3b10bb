#!/usr/bin/env python3
"""Idle time of the busiest queue in a rocprofv3 kernel trace of bench.py: for the second half of the run, every gap between two consecutive
kernels of that queue, summed by size class and by the pair of kernels around it, and the time during which NO queue ran a kernel.
usage: rocpd_gaps.py results.db total_steps [top]"""
import re
import sqlite3
import sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
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
nsteps = steps - steps // 2


def short(n):
    n = re.sub(r'^_ZN\d+_GLOBAL__N_1\d+', '', n)
    return re.sub(r'(I[LiEbf0-9]+E)?Ev.*|\.kd$', '', n)[:28]


busy = defaultdict(int)
for r in rows:
    busy[r[3]] += r[2] - r[1]
main = max(busy, key=busy.get)
mine = [r for r in rows if r[3] == main]
gaps = [(b[1] - a[2], short(a[0]), short(b[0])) for a, b in zip(mine, mine[1:]) if b[1] > a[2]]
tot = sum(g[0] for g in gaps)
print('main queue: %d kernels / step, busy %.2f ms / step, idle between its kernels %.2f ms / step' % (len(mine) / nsteps, busy[main] / nsteps / 1e6, tot / nsteps / 1e6))
for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 2e4), (2e4, 1e5), (1e5, 1e12)):
    sel = [g[0] for g in gaps if lo <= g[0] < hi]
    print('   gaps of %6.0f .. %8.0f ns: %6.1f / step, %.3f ms / step' % (lo, hi, len(sel) / nsteps, sum(sel) / nsteps / 1e6))
pair = defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    pair[(a, b)][0] += 1
    pair[(a, b)][1] += g
print('largest idle by (kernel before -> kernel after), per step:')
for (a, b), (n, t) in sorted(pair.items(), key=lambda kv: -kv[1][1])[:top]:
    print('   %-28s -> %-28s %6.1f x %7.1f us = %.3f ms' % (a, b, n / nsteps, t / n / 1e3, t / nsteps / 1e6))
# time with no kernel on any queue
ev = sorted([(r[1], 1) for r in rows] + [(r[2], -1) for r in rows])
depth, last, none = 0, ev[0][0], 0
for t, d in ev:
    if depth == 0:
        none += t - last
    depth += d
    last = t
print('no kernel on any queue: %.2f ms / step of a %.2f ms span' % (none / nsteps / 1e6, (rows[-1][2] - rows[0][1]) / nsteps / 1e6))

#!/usr/bin/env python3
"""The kernel sequence of ONE train step (the last complete one of a rocprofv3 kernel trace of bench.py), in start order, runs of the same kernel
collapsed: queue, short name, count, total microseconds -- where the many small launches of a step sit.  usage: rocpd_step_seq.py results.db total_steps"""
import re
import sqlite3
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
per = len(adam) // steps
rows = rows[adam[per * (steps - 2) - 1] + 1:adam[per * (steps - 1) - 1] + 1]
qs = {}
for r in rows:
    qs.setdefault(r[3], len(qs))


def short(n):
    n = re.sub(r'^_ZN\d+_GLOBAL__N_1\d+', '', n)
    n = re.sub(r'^_ZN2at6native\d*', 'at::', n)
    return re.sub(r'Ev.*|\.kd$', '', n)[:44]


out, t0 = [], rows[0][1]
for n, s, e, q in rows:
    k = (qs[q], short(n))
    if out and out[-1][0] == k:
        out[-1][1] += 1; out[-1][2] += e - s
    else:
        out.append([k, 1, e - s, s - t0])
for (q, n), c, d, at in out:
    print('%8.1f us  q%d  %-44s x%-3d %8.1f us' % (at / 1e3, q, n, c, d / 1e3))

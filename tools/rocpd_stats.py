#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share.
usage: rocpd_stats.py results.db [top_n] [skip_first_fraction]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in db.execute('pragma table_info(%s)' % kd)]
scol = [r[1] for r in db.execute('pragma table_info(%s)' % ks)]
name_col = 'kernel_name' if 'kernel_name' in scol else ('display_name' if 'display_name' in scol else scol[-1])
rows = db.execute('select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start' % (name_col, kd, ks)).fetchall()
if not rows:
    sys.exit('no dispatches')
t0, t1 = rows[0][1], max(r[2] for r in rows)
agg = {}
busy = 0
for n, s, e in rows:
    n = re.sub(r'\(.*', '', n)
    n = re.sub(r'^void ', '', n)
    a = agg.setdefault(n, [0, 0])
    a[0] += 1
    a[1] += e - s
    busy += e - s
print('dispatches %d, span %.1f ms, sum of kernel durations %.1f ms (%.0f%% of span)' % (len(rows), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0)))
print('%-90s %8s %12s %10s %7s' % ('kernel', 'calls', 'total_ms', 'avg_us', 'share'))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print('%-90s %8d %12.3f %10.1f %6.1f%%' % (n[:90], c, t / 1e6, t / c / 1e3, 100.0 * t / busy))

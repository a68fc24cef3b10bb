#!/usr/bin/env python3
"""Concurrency in a rocprofv3 kernel trace: per-queue busy time, union of busy intervals, time with >= 2 kernels in flight,
idle gaps.  usage: rocpd_overlap.py results.db [skip_first_fraction]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
cols = [r[1] for r in db.execute('pragma table_info(%s)' % kd)]
qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
rows = db.execute('select start, end, %s from %s order by start' % (qcol or '0', kd)).fetchall()
t0, t1 = rows[0][0], max(r[1] for r in rows)
cut = t0 + skip * (t1 - t0)
rows = [r for r in rows if r[0] >= cut]
span = max(r[1] for r in rows) - rows[0][0]
perq = {}
for s, e, q in rows:
    perq[q] = perq.get(q, 0) + (e - s)
ev = []
for s, e, _ in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = over = 0
depth, last = 0, ev[0][0]
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: over += t - last
    depth += d; last = t
print('span %.1f ms, kernels %d' % (span / 1e6, len(rows)))
for q, v in sorted(perq.items(), key=lambda kv: -kv[1]):
    print('  queue %s: sum of kernel durations %.1f ms (%.0f%% of span)' % (q, v / 1e6, 100.0 * v / span))
print('  GPU busy (>= 1 kernel in flight) %.1f ms = %.1f%% of span; >= 2 in flight %.1f ms; idle %.1f ms' % (
    busy / 1e6, 100.0 * busy / span, over / 1e6, (span - busy) / 1e6))

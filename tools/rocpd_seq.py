#!/usr/bin/env python3
"""Average duration of a kernel per consecutive GROUP of n calls in dispatch order (ablation scripts run n calls per variant).
usage: rocpd_seq.py results.db name-filter n"""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
flt, n = sys.argv[2], int(sys.argv[3])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
scol = [r[1] for r in db.execute('pragma table_info(%s)' % ks)]
name_col = 'kernel_name' if 'kernel_name' in scol else 'display_name'
rows = [(nm, e - s) for nm, s, e in db.execute('select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start' % (name_col, kd, ks)) if flt in nm]
for i in range(0, len(rows), n):
    g = rows[i:i + n]
    print('%-60s calls %3d  avg %.1f us' % (g[0][0][:60], len(g), sum(d for _, d in g) / len(g) / 1e3))

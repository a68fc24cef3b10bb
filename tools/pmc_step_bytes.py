#!/usr/bin/env python3
"""Whole-step HBM traffic from a rocprofv3 PMC database: sum of one counter (FETCH_SIZE or WRITE_SIZE, KiB) over every kernel dispatch, per
train step and per kernel family.   usage: pmc_step_bytes.py results.db COUNTER steps [top]
On gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section): the `bytes` column applies that
correction (read bytes = 2 * FETCH_SIZE * 1024); WRITE_SIZE is taken as KiB, uncorrected."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
counter, steps = sys.argv[2], int(sys.argv[3])
top = int(sys.argv[4]) if len(sys.argv) > 4 else 14
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]


def tab(prefix):
    return [t for t in tabs if t.startswith(prefix)][0]


kd, ks, pe, pi = tab('rocpd_kernel_dispatch'), tab('rocpd_info_kernel_symbol'), tab('rocpd_pmc_event'), tab('rocpd_info_pmc')


def cols(t):
    return [r[1] for r in db.execute('pragma table_info(%s)' % t)]


name_col = 'kernel_name' if 'kernel_name' in cols(ks) else 'display_name'
pname = 'name' if 'name' in cols(pi) else 'symbol'
q = ('select s.%s, e.value from %s e join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id join %s p on e.pmc_id = p.id '
     'where p.%s = ?' % (name_col, pe, kd, ks, pi, pname))
agg, total, n = {}, 0.0, 0
for kn, val in db.execute(q, (counter,)):
    kn = re.sub(r'\(.*', '', kn)
    kn = re.sub(r'^void ', '', kn)
    a = agg.setdefault(kn, [0, 0.0])
    a[0] += 1
    a[1] += val
    total += val
    n += 1
mult = 2048.0 if counter == 'FETCH_SIZE' else 1024.0
print('%s: %d dispatches over %d steps; per step %.1f MiB counted = %.3f GB of HBM traffic (x%d bytes per count)' % (
    counter, n, steps, total / steps / 1024.0, total * mult / steps / 1e9, int(mult)))
print('%-88s %8s %12s' % ('kernel', 'calls/st', 'GB/step'))
for kn, (c, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print('%-88s %8.1f %12.3f' % (kn[:88], c / steps, v * mult / steps / 1e9))

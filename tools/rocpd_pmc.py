#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database.  usage: rocpd_pmc.py results.db [name-filter]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ''
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
def tab(prefix):
    return [t for t in tabs if t.startswith(prefix)][0]
kd, ks, pe, pi = tab('rocpd_kernel_dispatch'), tab('rocpd_info_kernel_symbol'), tab('rocpd_pmc_event'), tab('rocpd_info_pmc')
def cols(t):
    return [r[1] for r in db.execute('pragma table_info(%s)' % t)]
if '--schema' in sys.argv:
    for t in (kd, pe, pi):
        print(t, cols(t))
    sys.exit(0)
scol = cols(ks)
name_col = 'kernel_name' if 'kernel_name' in scol else 'display_name'
pcol = cols(pi)
pname = 'name' if 'name' in pcol else 'symbol'
q = ('select s.%s, p.%s, e.value, d.end - d.start from %s e join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id '
     'join %s p on e.pmc_id = p.id' % (name_col, pname, pe, kd, ks, pi))
agg = {}
for kn, pn, val, dur in db.execute(q):
    kn = re.sub(r'\(.*', '', kn)
    if flt and flt not in kn:
        continue
    a = agg.setdefault((kn, pn), [0, 0.0, 0.0])
    a[0] += 1
    a[1] += val
    a[2] += dur
print('%-70s %-28s %8s %16s %12s' % ('kernel', 'counter', 'calls', 'avg value', 'avg dur us'))
for (kn, pn), (n, v, d) in sorted(agg.items(), key=lambda kv: -kv[1][2])[:80]:
    print('%-70s %-28s %8d %16.1f %12.1f' % (kn[:70], pn, n, v / n, d / n / 1e3))

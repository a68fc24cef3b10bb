"""How long the main stream waits at the side stream's join points in an eager step (HIP events around every SideStream.join and GradSink.join wait):
    python tools/side_join_wait.py [bench flags]"""
import contextlib
import io
import json
import sys
import torch
sys.path.insert(0, '.')
import bench  # noqa: E402
from ha2g_amd import ops  # noqa: E402

recs = []
_join = ops.SideStream.join


def join(self, device):
    if not self.enabled:
        return
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    _join(self, device)
    b.record()
    f = sys._getframe(1)
    recs.append((a, b, '%s:%d' % (f.f_code.co_name, f.f_lineno)))


ops.SideStream.join = join
sys.argv = [sys.argv[0], '--steps', '8', '--warmup', '4', '--no-cpu-baseline', '--primary-only', '--no-roofline', '--launch', 'eager'] + sys.argv[1:]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
torch.cuda.synchronize()
n = 12
by = {}
for a, b, who in recs:
    t = a.elapsed_time(b) * 1e3
    by.setdefault(who, []).append(t)
print('step %.3f ms (with the probe events); main-stream wait at side joins, per step:' % d['ms_per_step'])
tot = 0.0
for who, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print('   %-40s %5.1f joins x %7.1f us = %.3f ms' % (who, len(v) / n, sum(v) / len(v), sum(v) / n / 1e3))
    tot += sum(v) / n / 1e3
print('   total %.3f ms per step' % tot)

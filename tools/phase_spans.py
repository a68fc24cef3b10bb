"""Un-profiled phase spans of the eager train step: GPU events on the compute stream and host time stamps at the step's phase boundaries
(train_hierarchy.phase_clock), averaged over the timed steps of bench.py's primary leg.  Per phase: GPU span (event to event), host span (time the host spent
enqueueing it), and how far the host was AHEAD of the GPU when it finished enqueueing the phase (negative = the GPU was waiting for launches).
    python tools/phase_spans.py [bench flags]"""
import contextlib
import io
import json
import sys
sys.path.insert(0, '.')
import torch  # noqa: E402
import bench  # noqa: E402
from ha2g_amd import train_hierarchy as th  # noqa: E402

th.phase_clock = []
sys.argv = [sys.argv[0], '--steps', '12', '--warmup', '4', '--no-cpu-baseline', '--primary-only', '--no-roofline', '--launch', 'eager'] + sys.argv[1:]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
torch.cuda.synchronize()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
recs, th.phase_clock = th.phase_clock, None
steps = []
for r in recs:
    if r[0] == 'start':
        steps.append([])
    steps[-1].append(r)
steps = [s for s in steps if len(s) == len(steps[-1])][-12:]
print('eager step %.3f ms (%d steps analysed); host %.1f ms/step inside train_iter' % (d['ms_per_step'], len(steps), d['eager']['host_ms_per_step']))
print('%-52s %10s %10s' % ('phase', 'GPU ms', 'host ms'))
n = len(steps[0])
tot_g = tot_h = 0.0
for i in range(1, n):
    g = sum(s[i - 1][1].elapsed_time(s[i][1]) for s in steps) / len(steps)
    h = sum(s[i][2] - s[i - 1][2] for s in steps) / len(steps) * 1e3
    tot_g += g; tot_h += h
    print('%-52s %10.3f %10.3f' % (steps[0][i][0], g, h))
print('%-52s %10.3f %10.3f' % ('sum (start -> optimizers enqueued / executed)', tot_g, tot_h))
# step-to-step: GPU time between consecutive 'start' events = the step period on the device
per = [steps[k][0][1].elapsed_time(steps[k + 1][0][1]) for k in range(len(steps) - 1)]
print('device period between consecutive step starts: %.3f ms' % (sum(per) / len(per)))

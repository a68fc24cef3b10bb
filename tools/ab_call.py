"""bench.py's step with one library switch flipped first (A/B of debug entry points that have no environment variable):
    python tools/ab_call.py "lib.ha2g_conv_c32_prefetch(1 | 512)" [bench flags]      -> prints ms_per_step and value of the eager step"""
import contextlib
import io
import json
import sys
sys.path.insert(0, '.')
import bench  # noqa: E402
from ha2g_amd import ops, wav_engine as we  # noqa: E402,F401
from ha2g_amd._lib import lib  # noqa: E402,F401
stmt = sys.argv[1]
exec(stmt)
sys.argv = [sys.argv[0], '--steps', '12', '--warmup', '4', '--no-cpu-baseline', '--primary-only', '--no-roofline', '--launch', 'eager'] + sys.argv[2:]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print('%-48s %8.3f ms/step %10.1f' % (stmt, d['ms_per_step'], d['value']))

"""bf16-storage tower vs the oracle with the same rounding points, per tensor in network order (debug aid for tests/test_gpu_b16.py)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location('t', os.path.join(ROOT, 'tests', 'test_gpu_b16.py'))
m = importlib.util.module_from_spec(spec)
spec.loader.exec_module(m)
from ha2g_amd.config import CASES

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
case = dict(CASES['small'], B=B)
o32, o16, h16 = m._oracle_tower(case, False), m._oracle_tower(case, True), m._hip_tower(case, True)
h32 = m._hip_tower(case, False)
print('%-60s %9s %9s %9s %9s' % ('tensor', 'e_o', 'e_h', 'e_x', 'hip32'))
for k in o32:
    print('%-60s %9.2e %9.2e %9.2e %9.2e' % (k, m.rms_rel(o16[k], o32[k], o32[k]), m.rms_rel(h16[k], o32[k], o32[k]), m.rms_rel(h16[k], o16[k], o32[k]),
                                          m.rms_rel(h32[k], o32[k], o32[k])))

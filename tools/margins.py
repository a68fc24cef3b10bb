"""How close the GPU step is to the parity tolerance: runs the step / module parity tests in-process and prints the
largest error/tolerance ratios.  usage: [HA2G_GEMM_MODE=m] python tools/margins.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from ha2g_testing import Checker  # noqa: E402
import test_gpu_step as ts  # noqa: E402


def golden(name):
    return np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))


for name, fn in (('small', lambda: ts.test_train_step(golden, 'small', True)),
                 ('cfg1', lambda: ts.test_train_step(golden, 'cfg1', True)),
                 ('expr_small', lambda: ts.test_train_step_expressive(golden, True))):
    Checker.margins.clear()
    try:
        fn()
        status = 'pass'
    except AssertionError as e:
        status = 'FAIL ' + str(e)[:120]
    m = sorted(Checker.margins, reverse=True)
    r = np.array([x[0] for x in m])
    print('%-10s %s  comparisons %d  max err/tol %.3f  p99 %.3f  median %.4f' % (name, status, len(m), r[0], np.quantile(r, 0.99), np.median(r)))
    for ratio, key in m[:5]:
        print('     %.3f  %s' % (ratio, key))

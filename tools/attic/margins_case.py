"""Worst error/tolerance ratios of a whole-step parity case per matrix-core mode: python tools/margins_case.py expr_cfg1 6 0"""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from ha2g_testing import Checker
from ha2g_amd._lib import lib
import tests.test_gpu_step as T

name = sys.argv[1]
g = np.load('tests/golden/%s.npz' % name)
for mode in map(int, sys.argv[2:]):
    Checker.margins.clear()
    lib.ha2g_gemm_set_mode(mode)
    try:
        if 'expr' in name:
            T.test_train_step_expressive(lambda n: g, name, True)
        else:
            T.test_train_step(lambda n: g, name, True)
    except AssertionError as e:
        print('ASSERT', str(e)[:300])
    lib.ha2g_gemm_set_mode(6)
    m = sorted(Checker.margins, reverse=True)[:8]
    print('mode', mode, ' '.join('%.3f:%s' % (a, b.split('/', 2)[-1]) for a, b in m))

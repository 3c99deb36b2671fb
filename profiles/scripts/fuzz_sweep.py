"""One-off wider sweep of the randomised differential test (tests/test_gpu_fuzz.py) over seed
ranges the committed suite does not use:  python profiles/scripts/fuzz_sweep.py [first] [count]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest  # noqa: E402


class MP:
    def setenv(self, k, v):
        os.environ[k] = v


import __graft_entry__ as ge  # noqa: E402
import test_gpu_fuzz as tf  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad = 0
for lo in range(first, first + count, 50):
    for (nmax, mlo, mhi, sw) in ((600, 1, 13, None), (2500, 11, 33, None), (1200, 1, 25, "LBFGSB_TWO_PASS=0")):
        for k in ("LBFGSB_TWO_PASS",):
            os.environ.pop(k, None)
        try:
            tf.test_random_problems_against_oracle(po, MP(), lo, 50, nmax, mlo, mhi, sw)
            print("ok", lo, nmax, mlo, mhi, sw, flush=True)
        except AssertionError as e:
            bad += 1
            print("FAIL", lo, nmax, mlo, mhi, sw, str(e)[:400], flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)

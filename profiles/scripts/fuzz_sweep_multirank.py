"""One-off wider sweep of the sharded differential test (tests/test_gpu_multirank.py) on other
seeds:  python profiles/scripts/fuzz_sweep_multirank.py WORLD FIRST COUNT"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402,F401
import test_gpu_multirank as tm  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

world, first, count = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
bad = 0
for lo in range(first, first + count, 25):
    try:
        tm.test_sharded_random_problems_match_oracle(po, type("P", (), {"__truediv__": lambda s, o: os.path.join(tempfile.mkdtemp(), o)})(),
                                                     world, lo, 25)
        print("ok", world, lo, flush=True)
    except AssertionError as e:
        bad += 1
        print("FAIL", world, lo, str(e)[:400], flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)

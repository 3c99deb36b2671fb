"""One-off wider sweep of the randomised differential test (tests/test_gpu_fuzz.py: call by call beside the
oracle; every split must be reproduced by a one-step oracle replay) over seed ranges the committed suite does
not use, through both device-pointer entries, the fallback options and m > 32:

    python profiles/scripts/fuzz_sweep.py [first] [count] > gpurun_out/fuzz_sweep.txt
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as tf  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

po.build(ref=False)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
MODES = [  # (nmax, mlo, mhi, pp, options)
    (600, 1, 13, False, {}), (600, 1, 13, True, {}), (2500, 11, 33, True, {}), (2500, 11, 33, False, {}),
    (1200, 1, 25, True, {"two_pass": 0}), (1200, 1, 25, False, {"lean": 0}), (1200, 1, 25, True, {"uniform_bounds": 0}),
    (1200, 1, 25, True, {"spec_capture": 1}), (1000, 33, 90, True, {}), (1500, 1, 25, True, {"exact_always": 1}),
    # round 5: FEW-VALUED bounds (l, u drawn from <= 8 values each: dictionary-coded in the nbd byte), both entries,
    # and the same with the caller's arrays compared with the snapshot at every iteration
    (1500, 1, 25, True, {"_few_valued": 1}), (1500, 1, 25, False, {"_few_valued": 1, "bounds_check": 1}),
    # later in round 5: the previous forms of the second trial's evaluation and of the walk's first window
    (1500, 1, 25, True, {"spec_trial2": 0}), (1500, 1, 25, False, {"win_slack": 0}),
]
bad, total, splits, t0 = 0, 0, 0, time.time()
for seed in range(first, first + count):
    nmax, mlo, mhi, pp, opts = MODES[seed % len(MODES)]
    p = tf.make(po, seed, nmax, mlo, mhi)
    opts = dict(opts)
    if opts.pop("_few_valued", 0):
        import numpy as np
        rng = np.random.default_rng(seed)
        kl, ku = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        lv = np.sort(rng.normal(-1.5, 1.0, kl))
        uv = lv.max() + np.abs(rng.normal(1.0, 1.0, ku)) + 0.05
        p.l[:] = lv[rng.integers(0, kl, p.n)]
        p.u[:] = uv[rng.integers(0, ku, p.n)]
    try:
        split, _ = tf.drive_with_replay(po, p, 80, pp=pp, options=opts)
        total += 1
        splits += split is not None
    except AssertionError as e:
        bad += 1
        print("FAIL seed %d mode %s: %s" % (seed, MODES[seed % len(MODES)], str(e)[:600]), flush=True)
    if (seed - first) % 100 == 99:
        print("... %d problems, %d splits (each reproduced one-step), %d failures, %.0f s"
              % (total, splits, bad, time.time() - t0), flush=True)
print("problems %d  splits reproduced %d  (of them line-search branch flips at one ulp of g'd, "
      "tests/test_gpu_fuzz.py _line_search_branch_flip: %d %s)  failures %d  (%.0f s)"
      % (total, splits, len(tf.FLIPS), tf.FLIPS[:4], bad, time.time() - t0))
sys.exit(1 if bad else 0)

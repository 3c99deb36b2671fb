"""One-off sweep of the randomised differential test over problem FAMILIES the committed generator
(tests/test_gpu_fuzz.py: make) does not draw -- same harness (call by call beside the oracle; a split must be
reproduced by ONE oracle call from the GPU's previous state), both device-pointer entries:

  linear, scaled, sqrt, rosenchain, lattice, tiny  (tests/test_gpu_fuzz.py: FAMILIES; the committed suite runs
  25-60 seeds of each, this script any range)

    python profiles/scripts/fuzz_shapes.py [first] [count per family] > gpurun_out/fuzz_shapes.txt
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as tf  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

po.build(ref=False)


first = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 150
t0 = time.time()
worst = 0
for name, gen in tf.FAMILIES.items():
    bad = total = splits = nonfinite = drift = 0
    for seed in range(first, first + count):
        p = gen(po, seed)
        try:
            split, _ = tf.drive_with_replay(po, p, 60, pp=bool(seed & 1), final_check=False)
            total += 1
            splits += split is not None
            L = tf.LAST
            nonfinite += not (np.isfinite(L["f_oracle"]) and np.isfinite(L["f_gpu"]))
            if split is not None and np.isfinite(L["f_oracle"]) and np.isfinite(L["f_gpu"]):
                rel = abs(L["f_oracle"] - L["f_gpu"]) / max(1.0, abs(L["f_oracle"]))
                if rel > 1e-7:
                    # a split that one oracle call reproduced, two runs that then end apart: both cut at the
                    # iteration cap, or stopped by different (both legitimate) criteria
                    drift += 1
                    print("apart %s seed %d: split at call %d, f %.12g (%s) vs %.12g (%s), rel %.1e"
                          % (name, seed, split, L["f_oracle"], L["task_oracle"][:22], L["f_gpu"],
                             L["task_gpu"], rel), flush=True)
        except AssertionError as e:
            bad += 1
            print("FAIL %s seed %d: %s" % (name, seed, str(e)[:700]), flush=True)
    print("%-10s problems %d  splits reproduced one-step %d  (runs with NaN in the reference's own arithmetic %d, "
          "ended > 1e-7 apart after a reproduced split %d)  failures %d  (%.0f s)"
          % (name, total, splits, nonfinite, drift, bad, time.time() - t0), flush=True)
    worst += bad
sys.exit(1 if worst else 0)

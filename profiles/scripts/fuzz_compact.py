"""Wider sweep of the randomised differential test (tests/test_gpu_fuzz.py: call by call beside the oracle, every split
reproduced by a one-step oracle replay) with the option compact_w -- the two passes over W on the tile-local free-row
layout (m <= 10): the tiles re-sorted in every iteration and un-sorted by every export of the replay harness
(compact_w = 2), the automatic policy (compact_w = 1), both entries, the deferring contexts bit for bit against the
default ones, few-valued (dictionary-coded) and streamed bounds.

    python profiles/scripts/fuzz_compact.py [first] [count] > gpurun_out/fuzz_compact.txt
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import test_gpu_fuzz as tf  # noqa: E402
import test_gpu_defer as td  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

po.build(ref=False)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 400
C2 = {"compact_w": 2, "compact_policy": 2}
C1 = {"compact_w": 1, "compact_min_rows": 0}
MODES = [  # (nmax, mlo, mhi, pp, options, kind)
    (600, 1, 11, False, C2, "replay"), (600, 1, 11, True, C2, "replay"), (3000, 3, 11, True, C2, "replay"),
    (3000, 3, 11, False, C1, "replay"), (3000, 3, 11, True, C1, "replay"),
    (1500, 1, 11, True, dict(C2, uniform_bounds=0), "replay"), (1500, 1, 11, True, dict(C2, _few_valued=1), "replay"),
    (1500, 1, 11, False, dict(C2, lean=0), "replay"), (1500, 1, 11, True, dict(C2, two_pass=0), "replay"),
    (2000, 1, 11, True, C2, "defer"), (2000, 1, 11, False, C1, "defer"),
]
bad, total, splits, t0 = 0, 0, 0, time.time()
for seed in range(first, first + count):
    nmax, mlo, mhi, pp, opts, kind = MODES[seed % len(MODES)]
    p = tf.make(po, seed, nmax, mlo, mhi)
    opts = dict(opts)
    if opts.pop("_few_valued", 0):
        rng = np.random.default_rng(seed)
        kl, ku = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        lv = np.sort(rng.normal(-1.5, 1.0, kl))
        uv = lv.max() + np.abs(rng.normal(1.0, 1.0, ku)) + 0.05
        p.l[:] = lv[rng.integers(0, kl, p.n)]
        p.u[:] = uv[rng.integers(0, ku, p.n)]
    try:
        if kind == "replay":
            split, _ = tf.drive_with_replay(po, p, 80, pp=pp, options=opts)
            splits += split is not None
        else:
            td._same(p, pp, max_iter=60, options=opts)
        total += 1
    except AssertionError as e:
        bad += 1
        print("FAIL seed %d mode %s: %s" % (seed, MODES[seed % len(MODES)], str(e)[:600]), flush=True)
    if (seed - first) % 100 == 99:
        print("... %d problems, %d splits (each reproduced one-step), %d failures, %.0f s"
              % (total, splits, bad, time.time() - t0), flush=True)
print("problems %d  splits reproduced %d  failures %d  (%.0f s)" % (total, splits, bad, time.time() - t0))
sys.exit(1 if bad else 0)

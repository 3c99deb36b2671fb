"""Wider sweep of tests/test_gpu_defer.py: LBFGSB_F_DEFER_LNSRCH against the default mode, BIT FOR BIT at every
NEW_X return, the final return and the exported state, over seed ranges and problem families the committed suite
does not use, both device-pointer entries, m up to 32, the fallback options:

    python profiles/scripts/defer_sweep.py [first] [count] > gpurun_out/defer_sweep.txt
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as tf      # noqa: E402
import test_gpu_defer as td     # noqa: E402
from test_gpu_parity import _random_box_rosenbrock   # noqa: E402
from oracle import pyoracle as po  # noqa: E402

po.build(ref=False)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
GENS = [
    lambda s: tf.make(po, s, 600, 1, 13), lambda s: tf.make(po, s, 2500, 11, 33), lambda s: _random_box_rosenbrock(po, s),
    lambda s: tf.fam_linear(po, s), lambda s: tf.fam_lattice(po, s), lambda s: tf.fam_rosenchain(po, s),
    lambda s: tf.fam_scaled(po, s), lambda s: tf.fam_sqrt(po, s), lambda s: tf.fam_tiny(po, s),
    lambda s: tf.make(po, s, 1200, 1, 25),
    # m > 32: the one-launch r pass defers its sums too; and the tiled launches (which wait for theirs) beside it
    lambda s: tf.make(po, s, 1200, 33, 90), lambda s: tf.make(po, s, 1200, 33, 90),
]
OPTS = [{}, {}, {}, {}, {}, {}, {}, {}, {}, {"two_pass": 0}, {}, {"wide_one": 0}]
bad, total, deferred, reissued, t0 = 0, 0, 0, 0, time.time()
for seed in range(first, first + count):
    k = seed % len(GENS)
    p = GENS[k](seed)
    pp = (seed // len(GENS)) % 2 == 0
    try:
        d, r = td._same(p, pp, max_iter=120, **({"options": OPTS[k]} if OPTS[k] else {}))
        total += 1
        deferred += d
        reissued += r
    except AssertionError as e:
        bad += 1
        print("FAIL seed %d gen %d pp %s: %s" % (seed, k, pp, str(e)[:500]), flush=True)
    if (seed - first) % 100 == 99:
        print("... %d problems, %d set-ups deferred, %d requests re-issued, %d failures, %.0f s"
              % (total, deferred, reissued, bad, time.time() - t0), flush=True)
print("problems %d  set-ups deferred %d  requests re-issued %d  failures %d  (%.0f s)"
      % (total, deferred, reissued, bad, time.time() - t0))
sys.exit(1 if bad else 0)

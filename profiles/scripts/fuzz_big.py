"""One-off: the one-step-replay harness (tests/test_gpu_fuzz.py) on LARGE random problems -- n up to 6e5, where the
breakpoint provider works in windows, refills, merges and keeps the rows a walk fixes as a cursor instead of a
list -- from the general generator and from the families with long walks (linear) and huge tie groups (lattice).

    python profiles/scripts/fuzz_big.py [first] [count per kind] [kind substring] > gpurun_out/fuzz_big.txt
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
first = int(sys.argv[1]) if len(sys.argv) > 1 else 97000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 8


KINDS = [("make x300", lambda po_, s: tf.scaled_up(lambda q, t: tf.make(q, t, 2000, 1, 9), 300)(po_, s)),
         ("linear x300", tf.scaled_up(tf.FAMILIES["linear"], 300)),
         ("lattice x400", tf.scaled_up(tf.FAMILIES["lattice"], 400)),
         ("rosenchain x500", tf.scaled_up(tf.FAMILIES["rosenchain"], 500)),
         # 21 ... 32 pairs (the update pass split over the columns), more than 32, skipped updates
         ("make m21-32 x150", lambda po_, s: tf.scaled_up(lambda q, t: tf.make(q, t, 2000, 21, 33), 150)(po_, s)),
         ("make m33-60 x100", lambda po_, s: tf.scaled_up(lambda q, t: tf.make(q, t, 1500, 33, 61), 100)(po_, s)),
         ("cubic x300", tf.scaled_up(tf.FAMILIES["cubic"], 300))]
if len(sys.argv) > 3:   # a substring selects kinds
    KINDS = [k for k in KINDS if sys.argv[3] in k[0]]
t0 = time.time()
worst = 0
for name, gen in KINDS:
    bad = tot = spl = 0
    ns = []
    for seed in range(first, first + count):
        p = gen(po, seed)
        ns.append(p.n)
        try:
            # (long enough for the memory to fill where m is large)
            s, _ = tf.drive_with_replay(po, p, 25 if p.m < 20 else p.m + 25, pp=bool(seed & 1), final_check=False)
            tot += 1
            spl += s is not None
        except AssertionError as e:
            bad += 1
            print("FAIL %s seed %d (n=%d m=%d): %s" % (name, seed, p.n, p.m, str(e)[:600]), flush=True)
        print("  .. %s seed %d n=%d m=%d  %.0f s" % (name, seed, p.n, p.m, time.time() - t0), flush=True)
    print("%-16s problems %d (n = %d … %d)  splits reproduced one-step %d  failures %d  (%.0f s)"
          % (name, tot, min(ns), max(ns), spl, bad, time.time() - t0), flush=True)
    worst += bad
sys.exit(1 if worst else 0)

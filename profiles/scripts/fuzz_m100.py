"""m = 97 ... 130: beyond the one-launch r pass (WIDE_MAXC = 96 columns): one launch per tile of 32 columns behind
pair_commit, the line-search sums waited for in the same call.  python profiles/scripts/fuzz_m100.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as tf
from oracle import pyoracle as po
po.build(ref=False)
bad = 0
for seed in range(140000, 140040):
    p = tf.make(po, seed, 900, 97, 130)
    try:
        tf.drive_with_replay(po, p, 150, pp=seed % 2 == 0, options={})
    except AssertionError as e:
        bad += 1
        print("FAIL", seed, str(e)[:400], flush=True)
print("m = 97...130: problems 40 failures", bad)

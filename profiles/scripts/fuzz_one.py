"""One problem of the randomised differential test under several option sets (what a failure of a sweep is narrowed
down with):   python profiles/scripts/fuzz_one.py SEED NMAX MLO MHI PP 'json options' ['json options' ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as tf  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

po.build(ref=False)
seed, nmax, mlo, mhi, pp = (int(v) for v in sys.argv[1:6])
for js in sys.argv[6:]:
    opts = json.loads(js)
    p = tf.make(po, seed, nmax, mlo, mhi)
    try:
        split, _ = tf.drive_with_replay(po, p, 80, pp=bool(pp), options=opts)
        print("options %s: ok (n = %d, m = %d, split %s)" % (js, p.n, p.m, split), flush=True)
    except AssertionError as e:
        print("options %s: FAIL %s" % (js, str(e)[:1500]), flush=True)

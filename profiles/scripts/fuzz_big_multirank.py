"""One-off: LARGE random problems of several families SHARDED over 2-4 ranks on one GPU, through the library's
communicator code path (tests/fake_rccl.cpp stands in for librccl) or the gloo host reducers, under the
one-step-replay bar (tests/_mr_worker.py: run_fuzz; the first call that leaves the single-rank oracle's trajectory
must be reproduced by one oracle call from the assembled state of the ranks).

    python profiles/scripts/fuzz_big_multirank.py WORLD FIRST COUNT [fakerccl|gloo] > gpurun_out/fuzz_big_mr.txt
"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402,F401
import test_gpu_multirank as tm  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

po.build(ref=False)
world, first, count = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
comm = sys.argv[4] if len(sys.argv) > 4 else "fakerccl"
if comm == "fakerccl":
    os.environ["LBFGSB_RCCL_LIBRARY"] = tm._fake_rccl()
KINDS = [("make", 200), ("linear", 200), ("lattice", 300), ("rosenchain", 300), ("scaled", 100)]
t0 = time.time()
worst = 0
for fam, scale in KINDS:
    out = os.path.join(tempfile.mkdtemp(), "out.json")
    res = tm.launch(world, "fuzz:%s:%d:%s" % (fam, scale, comm), first, count, 25, "-", out)
    bad = spl = 0
    for seed, r in res.items():
        if r["split"] is None:
            if r["calls"] != r["oracle_calls"]:
                bad += 1
                print("FAIL %s seed %s: %d calls vs %d" % (fam, seed, r["calls"], r["oracle_calls"]), flush=True)
        else:
            spl += 1
            if r["verdict"] != "reproduced":
                bad += 1
                print("FAIL %s seed %s: %s" % (fam, seed, r["verdict"][:600]), flush=True)
    print("%-12s x%d over %d ranks (%s): problems %d  splits reproduced one-step %d  failures %d  (%.0f s)"
          % (fam, scale, world, comm, len(res), spl, bad, time.time() - t0), flush=True)
    worst += bad
sys.exit(1 if worst else 0)

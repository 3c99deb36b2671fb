#!/usr/bin/env python3
"""Per-iteration timeline of the headline run (n = 1e8, m = 10, fp64, ping-pong entry + LBFGSB_F_DEFER_LNSRCH,
deferred f): wall ms between NEW_X returns, nfg, nseg, nfree, host syncs and kernel launches of each iteration --
where the mean of an iteration differs from its median.

    python profiles/scripts/iter_timeline.py [--n ROWS] [--m M] [--iters K] > profiles/roundN_iter_timeline.txt
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100_000_000)
    ap.add_argument("--m", type=int, default=10)
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--opt", action="append", default=[])
    a = ap.parse_args()
    import torch
    import lbfgsb_amd as la
    opts = {k: float(v) for k, v in (kv.split("=", 1) for kv in a.opt)}
    sol = la.DeviceSolver(a.n, a.m, same_stream_objective=True, defer_lnsrch=True, options=opts)
    x = torch.zeros(a.n, dtype=torch.float64, device="cuda")
    xs, gs = [x, torch.empty_like(x)], [torch.zeros_like(x), torch.empty_like(x)]
    l, u = torch.full_like(x, -1.0), torch.full_like(x, 1.0)
    nbd = torch.full((a.n,), 2, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    t_prev = time.perf_counter()
    st_prev = sol.stats()
    print("iter  nfg      nseg      nfree   ms     syncs launches")
    cur = 0
    while True:
        t, cur = sol.setulb_pp(xs, l, u, nbd, gs, 0.0, 0.0)
        if t.startswith("FG"):
            sol.objective(0, xs[cur], gs[cur], deferred=True)
        elif t.startswith("NEW_X"):
            now = time.perf_counter()
            st = sol.stats()
            print("%4d %4d %9d %10d %7.3f %5d %5d" % (sol.isave[29], sol.isave[33], sol.isave[32], sol.isave[37],
                                                      (now - t_prev) * 1e3, st["syncs"] - st_prev["syncs"],
                                                      st["launches"] - st_prev["launches"]), flush=True)
            t_prev, st_prev = now, st
            if sol.isave[29] >= a.iters:
                break
        else:
            print("ended:", t)
            break
    sol.close()


if __name__ == "__main__":
    main()

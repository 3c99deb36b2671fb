#!/usr/bin/env python3
"""host syncs per iteration of the same problem (separable bounded quadratic, per-variable bounds) driven through the
ping-pong entry + LBFGSB_F_DEFER_LNSRCH by (a) the library's objective on the solver's stream, (b) a torch objective on
torch's stream ordered with events (stream_ordered) and f as a device scalar.  Usage: python torch_caller_syncs.py [n]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import lbfgsb_amd as la
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
m = 10
dev = torch.device("cuda", 0)
for mode in ("library objective, same stream", "torch objective, stream_ordered"):
    tq = mode.startswith("torch")
    sol = la.DeviceSolver(n, m, same_stream_objective=not tq, stream_ordered=tq, defer_lnsrch=True,
                          options={"compact_w": 1})
    x, l, u, nbd = bench.problem_tensors(torch, dev, 0, n, 0, False, True)
    xs, gs = [x, torch.empty_like(x)], [torch.zeros_like(x), torch.empty_like(x)]
    if tq:
        i = torch.arange(1, n + 1, dtype=torch.int64, device=dev)
        a_ = 1.0 + 99.0 * ((7919 * i) % 10007).to(torch.float64) / 10006.0
        c_ = -2.0 + 4.0 * ((104729 * i) % 100003).to(torch.float64) / 100002.0
        tmp = torch.empty_like(x)
    it0 = None
    rows = []
    while True:
        t, cur = sol.setulb_pp(xs, l, u, nbd, gs, 0.0, 0.0)
        if t.startswith("FG"):
            if tq:
                d = torch.sub(xs[cur], c_, out=tmp)
                torch.mul(a_, d, out=gs[cur])
                sol.set_f_device(0.5 * torch.dot(gs[cur], d))
            else:
                sol.objective(0, xs[cur], gs[cur], deferred=True)
        elif t.startswith("NEW_X"):
            it = int(sol.isave[29])
            rows.append((it, int(sol.isave[33]), int(sol.isave[32]), int(sol.isave[37]), float(sol.f[0])))
            if it == 12:
                torch.cuda.synchronize()
                it0, s0, t0 = it, sol.stats(), time.perf_counter()
            if it == 42:
                torch.cuda.synchronize()
                s1, t1 = sol.stats(), time.perf_counter()
                break
        else:
            raise SystemExit(t)
    k = 42 - it0
    print("%-34s n=%d: %.2f syncs/iter, %.2f launches/iter, %.3f ms/iter, deferred %s, last row %s"
          % (mode, n, (s1["syncs"] - s0["syncs"]) / k, (s1["launches"] - s0["launches"]) / k, (t1 - t0) / k * 1e3,
             sol.defer_stats(), rows[-1]))
    sol.close()

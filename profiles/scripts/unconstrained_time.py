"""Iteration rate on an UNCONSTRAINED problem (nbd = 0: mainlb :607-611 skips the Cauchy search):
the separable quadratic of the bench without its bounds, n = 5e7, m = 10, fp64.
   python profiles/scripts/unconstrained_time.py            (two passes over W per iteration)
   LBFGSB_TWO_PASS=0 python profiles/scripts/unconstrained_time.py   (update_pairs + cmprlb + subsm)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import lbfgsb_amd

n, m = int(os.environ.get("N", 50_000_000)), 10
sol = lbfgsb_amd.DeviceSolver(n, m, same_stream_objective=True)
x = torch.zeros(n, dtype=torch.float64, device="cuda")
g = torch.zeros_like(x)
l, u = torch.full_like(x, -1.0), torch.full_like(x, 1.0)
nbd = torch.zeros(n, dtype=torch.int32, device="cuda")
it, t0, marks = 0, None, []
while it < 40:
    t = sol.setulb(x, l, u, nbd, g, 0.0, 0.0)
    if t.startswith("FG"):
        sol.f[0] = sol.objective(0, x, g)
    elif t.startswith("NEW_X"):
        it += 1
        torch.cuda.synchronize()
        marks.append(time.perf_counter())
    else:
        break
dt = (marks[-1] - marks[14]) / (len(marks) - 15)
print("unconstrained n=%d m=%d: %.3f ms per iteration = %.1f it/s; task %s; path counts %s" %
      (n, m, dt * 1e3, 1.0 / dt, sol.task_s[:20], sol.path_counts()))
sol.close()

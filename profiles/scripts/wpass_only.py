"""Run a few iterations at the bench size (so that col = m and l, u, nbd are known to the
context), then launch each of the three W passes of an iteration a few times, for PMC passes:
   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python3 profiles/scripts/wpass_only.py
   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT -- python3 profiles/scripts/wpass_only.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import lbfgsb_amd

n = int(os.environ.get("WTV_N", "100000000"))
m = int(os.environ.get("WTV_M", "10"))
reps = int(os.environ.get("WTV_REPS", "3"))
sol = lbfgsb_amd.DeviceSolver(n, m, same_stream_objective=True, parallel_gcp=True)
x = torch.zeros(n, dtype=torch.float64, device="cuda")
g = torch.zeros_like(x)
l, u = torch.full_like(x, -1.0), torch.full_like(x, 1.0)
nbd = torch.full((n,), 2, dtype=torch.int32, device="cuda")
it = 0
while it < m + 2:
    t = sol.setulb(x, l, u, nbd, g, 0.0, 0.0)
    if t.startswith("FG"):
        sol.f[0] = sol.objective(0, x, g)
    elif t.startswith("NEW_X"):
        it += 1
    else:
        break
torch.cuda.synchronize()
col, head = int(sol.isave[27]), int(sol.isave[26])
for which in (2, 4, 3):
    print("which", which, "col", col, "avg ms per launch (hipEvents):",
          sol.kernel_time(which, x, g, col, head, reps))
sol.close()

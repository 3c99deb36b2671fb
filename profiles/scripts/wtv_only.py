"""Launch only the WS/WY matvec kernel (wtv_kernel) at the bench size, for PMC passes:
   rocprofv3 --pmc FETCH_SIZE  --kernel-trace --output-format csv -d OUT -- python3 profiles/scripts/wtv_only.py
   rocprofv3 --pmc WRITE_SIZE  --kernel-trace --output-format csv -d OUT -- python3 profiles/scripts/wtv_only.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import lbfgsb_amd

n = int(os.environ.get("WTV_N", "100000000"))
m = int(os.environ.get("WTV_M", "10"))
reps = int(os.environ.get("WTV_REPS", "5"))
sol = lbfgsb_amd.DeviceSolver(n, m)
v = torch.randn(n, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for _ in range(reps):
    sol.wtv_launch(v, m, 1)
sol.sync()
print("avg ms per launch (hipEvents):", sol.wtv_time(v, m, 1, reps))
sol.close()

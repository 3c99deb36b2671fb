"""Launch only cmprlb_wtv_kernel (r of cmprlb + W'r of subsm) at the bench size, for PMC passes:
   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python3 profiles/scripts/cmprlb_wtv_only.py
   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT -- python3 profiles/scripts/cmprlb_wtv_only.py
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
x = torch.randn(n, dtype=torch.float64, device="cuda")
g = torch.randn(n, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
# which = 2: the variant the iteration runs (formk new-row sums riding along)
print("avg ms per launch (hipEvents):", sol.kernel_time(2, x, g, m, 1, reps))
sol.close()

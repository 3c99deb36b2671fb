"""Time of the from-scratch formk Gram pass (formk_gram_rows_kernel) at n = 1e8, m = 10, fp64:
   python profiles/scripts/gram_time.py  ->  ms per launch, GB/s of the algorithmic bytes
   ((2 col) n s + n for iwhere)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lbfgsb_amd  # noqa: E402

n, m = int(os.environ.get("N", 100_000_000)), int(os.environ.get("M", 10))
for real32 in (False, True):
    sol = lbfgsb_amd.DeviceSolver(n, m, real32=real32)
    dt = torch.float32 if real32 else torch.float64
    x = torch.zeros(n, dtype=dt, device="cuda")
    g = torch.ones(n, dtype=dt, device="cuda")
    torch.cuda.synchronize()
    ms = sol.kernel_time(1, x, g, m, 1, 10)
    by = 2 * m * n * (4 if real32 else 8) + n
    print("formk_gram_rows_kernel<%s, %d>: %.3f ms  %.0f GB/s  (%.0f%% of 8 TB/s)"
          % ("float" if real32 else "double", m, ms, by / ms / 1e6, by / ms / 1e6 / 80))
    sol.close()
    del x, g
    torch.cuda.empty_cache()

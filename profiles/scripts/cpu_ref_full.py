#!/usr/bin/env python3
"""One full-size run of the untouched reference (oracle/_ref, -fdefault-integer-8 build) on the GPU
box's host: separable bounded quadratic, n = 1e8, m = 10, fp64.  Times every
setulb call (objective excluded), prints one JSON object with the first-iteration seconds, the
seconds per iteration once col = m, and the peak RSS.  bench.py's cpu_baseline leg times n = 2e7 and
scales x5; this is the measurement that scaling is checked against.

    python profiles/scripts/cpu_ref_full.py [--n 100000000] [--m 10] [--iters 13] > gpurun_out/cpu_ref_full.json

Test/measurement infrastructure only (it loads oracle/); nothing in the product imports it.
"""
import argparse
import json
import os
import resource
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def mem_available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) / 1048576.0
    except OSError:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100_000_000)
    ap.add_argument("--m", type=int, default=10)
    ap.add_argument("--iters", type=int, default=13)
    ap.add_argument("--engine", default="ref_i8")
    ap.add_argument("--budget-s", type=float, default=900.0)
    ap.add_argument("--mem-factor", type=float, default=1.5,
                    help="refuse to start unless MemAvailable >= this x the arrays' size (m = 20 at n = 1e8 needs "
                         "42 GB: 1.3 on a 62 GB host)")
    a = ap.parse_args()
    from bench import host_cpu
    from oracle import pyoracle as po

    real_bytes = 4 if a.engine.endswith("r32") or "r32" in a.engine else 8
    need_gb = ((2 * a.m + 5 + 4) * a.n * real_bytes + 4 * a.n * 8) / 2**30
    avail = mem_available_gb()
    if avail is not None and avail < a.mem_factor * need_gb:
        print(json.dumps({"error": "not enough host memory", "need_gb": need_gb, "available_gb": avail}))
        return 1
    eng = po.Engine(a.engine)
    real = eng.real
    p = po.problem_quadratic(a.n, a.m, real=real)
    s = po.State.fresh(p, eng.int)
    t_in, marks, rows = 0.0, [], []
    t_start = time.time()
    while True:
        t0 = time.perf_counter()
        po.call(eng, p, s)
        t_in += time.perf_counter() - t0
        t = s.task_s
        if t.startswith("FG"):
            s.f[0] = p.fg(s.x, s.g)
        elif t.startswith("NEW_X"):
            marks.append(t_in)
            rows.append({"iter": int(s.isave[29]), "nfg": int(s.isave[33]), "nseg": int(s.isave[32]),
                         "nfree": int(s.isave[37]), "col": int(s.isave[27]), "f": float(s.f[0]),
                         "sbgnrm": float(s.dsave[12]), "t_setulb_cum_s": t_in})
            sys.stderr.write("iter %d  col %d  nseg %d  nfree %d  f %.16e  t_in %.2f s\n"
                             % (rows[-1]["iter"], rows[-1]["col"], rows[-1]["nseg"], rows[-1]["nfree"],
                                rows[-1]["f"], t_in))
            sys.stderr.flush()
            if len(marks) >= a.iters or time.time() - t_start > a.budget_s:
                break
        else:
            break
    full = [k for k, r in enumerate(rows) if r["col"] == a.m]
    per_iter = None
    if len(full) >= 2:
        per_iter = (marks[full[-1]] - marks[full[0]]) / (full[-1] - full[0])
    cpu = host_cpu()
    out = {"engine": a.engine, "n": a.n, "m": a.m, "iterations_run": len(rows),
           "first_iteration_s": marks[0] if marks else None,
           "s_per_iter_col_eq_m": per_iter,
           "iters_per_sec_col_eq_m": (1.0 / per_iter) if per_iter else None,
           "iters_timed_col_eq_m": (full[-1] - full[0]) if len(full) >= 2 else 0,
           "peak_rss_gb": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0,
           "final_task": s.task_s, "host": cpu, "cores_used": 1, "rows": rows,
           "wall_s": time.time() - t_start}
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())

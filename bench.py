#!/usr/bin/env python3
"""bench.py -- setulb iterations/sec on the MI355X-native L-BFGS-B inner iteration.

Workload (BASELINE.json metric): separable bounded quadratic, n = 1e8, m = 10, fp64,
l = -1, u = +1, x0 = 0, factr = pgtol = 0, on-device objective.  A "step"
is one L-BFGS-B iteration (one NEW_X return): every kernel of the hot path runs, plus the
f/g evaluations the line search asks for.  With --gpus N the n rows are sharded over N
ranks (STRONG scaling: n stays 1e8) and every reduction is completed with RCCL all-reduces
on the <= 4m+5 partials.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n ROWS] [--m M]

Prints ONE JSON line (rank 0).  `value` = K / wall time of the K timed iterations
(objective evaluations included; `iters_per_sec_setulb_only` excludes them).
`roofline` is measured live on the WS/WY matvec kernel (wtv_kernel) with HIP events on the
stream it runs on.  `cpu_baseline` times the real reference (oracle/_ref) on host cores,
rank 0 at N=1 only, on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# the host driver of this pool only supports dmabuf IPC: RCCL needs this for multi-process runs
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--n", "--rows", dest="n", type=int, default=100_000_000,
                    help="global number of variables (--rows: spelling that torch.distributed.run's "
                         "own parser does not mistake for one of its options)")
    ap.add_argument("--m", type=int, default=10)
    ap.add_argument("--real32", action="store_true", help="REAL32 context (BASELINE.json configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=20_000_000,
                    help="rows of the CPU reference sample (same problem, same m)")
    ap.add_argument("--rccl-self", action="store_true",
                    help="single GPU: attach a 1-rank RCCL communicator, so that every reduction pays a "
                         "real ncclAllReduce + D2H + sync (latency floor of the sharded path; use with "
                         "--rows 12500000 = the per-rank shape of n=1e8 over 8 GPUs)")
    ap.add_argument("--allow-gloo-fallback", action="store_true",
                    help="multi-GPU: if the RCCL communicator cannot be created, complete the reductions "
                         "through a gloo host group instead of exiting non-zero")
    ap.add_argument("--roofline-reps", type=int, default=20)
    return ap.parse_args()


class stdout_to_stderr:
    """RCCL prints a version banner on STDOUT when a communicator is created; this file must print
    exactly one JSON line there.  Inside the block file descriptor 1 points at stderr."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def host_cpu():
    """model name / core counts of the host this process runs on"""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count()
    return {"model": model, "cores_total": os.cpu_count(), "cores_usable": usable}


def cpu_baseline(m, n_full, n_sample):
    """The untouched reference (oracle/_ref, amdflang -O2; the -fdefault-integer-8 build, which is
    the one that can run n = 1e8 at all) on ONE host core -- it is single
    threaded by construction -- on the same problem at n_sample rows.  Iterations with col = m
    are timed inside setulb only (the objective is excluded); the value is scaled linearly in n
    to n_full rows (every loop of the reference is O(n) at fixed m; SURVEY.md section 6 measured
    0.19 / 1.95 / 21.8 s per iteration at n = 1e6 / 1e7 / 1e8)."""
    from oracle import pyoracle as po
    kind, eng = "reference", None
    for name in ("ref_i8", "ref"):
        try:
            eng = po.Engine(name)
            break
        except (FileNotFoundError, OSError):
            continue
    if eng is None:
        eng = po.Engine("oracle")
        kind = "port"
    p = po.problem_quadratic(n_sample, m)
    s = po.State.fresh(p, eng.int)
    t_in, marks, cols = 0.0, [], []
    timed = 3                        # iterations timed at col = m
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
            cols.append(int(s.isave[27]))
            # col at NEW_X k is the col of iteration k's own work; iteration k+1 runs matupd first
            full = [k for k, c in enumerate(cols) if c == m]
            if len(full) >= timed + 1 or time.time() - t_start > 240:
                break
        else:
            break
    full = [k for k, c in enumerate(cols) if c == m]
    if len(full) >= 2:
        k0, k1 = full[0], full[-1]       # iterations k0+1 .. k1 ran entirely with col = m
    else:
        k0, k1 = max(0, len(marks) - 3), len(marks) - 1
    k = max(1, k1 - k0)
    per_iter = (marks[k1] - marks[k0]) / k
    cpu = host_cpu()
    return {
        "value": 1.0 / per_iter * n_sample / n_full,
        "unit": "iters/sec",
        "cores": 1,
        "kind": kind,
        "engine": eng.kind,
        "n_sample": n_sample,
        "s_per_iter_at_sample": per_iter,
        "iters_timed": k,
        "host_cpu_model": cpu["model"],
        "host_cores_total": cpu["cores_total"],
        "host_cores_usable": cpu["cores_usable"],
        "sample": "n=%d rows (1/%g of the workload), m=%d, %d iterations with col=m (of %d run), "
                  "time inside setulb only: %.4f s/iter at the sample size; value scaled linearly "
                  "in n to n=%d; first iteration (nseg~0.977n) took %.2f s; %s on 1 of %d cores "
                  "(the reference is single-threaded)"
                  % (n_sample, n_full / n_sample, m, k, len(marks), per_iter, n_full, marks[0],
                     cpu["model"], cpu["cores_total"]),
    }


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    import lbfgsb_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % a.gpus)
    # LBFGSB_BENCH_SHARE_GPU=1 (tests): every rank on cuda:0 with a gloo group, so that this
    # file's multi-rank flow can be rehearsed on a one-GPU box (with LBFGSB_RCCL_LIBRARY pointing
    # the library at the shared-memory stand-in of tests/fake_rccl.cpp)
    share_gpu = os.environ.get("LBFGSB_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        with stdout_to_stderr():
            if share_gpu:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)

    n, m = a.n, a.m
    # contiguous block sharding of the rows
    row0, n_loc = lbfgsb_amd.block_partition(n, world, rank)
    # the objective below is the library's own kernel on the solver's stream, so the FG return
    # needs no host sync
    sol = lbfgsb_amd.DeviceSolver(n_loc, m, n_global=n, row0=row0, device=local_rank,
                                  same_stream_objective=True, real32=a.real32)
    rdt = torch.float32 if a.real32 else torch.float64
    rbytes = 4 if a.real32 else 8
    collective = "none"
    if world > 1:
        # RCCL on the solver's stream.  A scaling curve must never silently be a gloo curve: if
        # the communicator cannot be created on some rank the run exits non-zero, unless
        # --allow-gloo-fallback asks for the (tiny) reductions to go through a gloo host group.
        ok = 1
        with stdout_to_stderr():
            try:
                lbfgsb_amd.attach_rccl(sol, rank, world, dev)
            except Exception as e:   # noqa: BLE001
                ok = 0
                sys.stderr.write("rank %d: RCCL communicator failed (%r)\n" % (rank, e))
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            collective = "RCCL all-reduce of <=4m+11 fp64 partials per phase"
        elif not a.allow_gloo_fallback:
            sol.close()
            dist.destroy_process_group()
            raise SystemExit("RCCL communicator unavailable on some rank (pass --allow-gloo-fallback "
                             "to measure with a gloo host group instead)")
        else:
            sol.close()
            sol = lbfgsb_amd.DeviceSolver(n_loc, m, n_global=n, row0=row0, device=local_rank,
                                          same_stream_objective=True, real32=a.real32)
            lbfgsb_amd.attach_host_group(sol, rank, world, group=dist.new_group(backend="gloo"))
            collective = "gloo host all-reduce (RCCL communicator unavailable; --allow-gloo-fallback)"
    elif a.rccl_self:
        with stdout_to_stderr():
            lbfgsb_amd.attach_rccl(sol, 0, 1, dev)
        collective = "RCCL all-reduce on a 1-rank communicator (--rccl-self: latency floor)"

    x = torch.zeros(n_loc, dtype=rdt, device=dev)
    g = torch.zeros_like(x)
    l = torch.full_like(x, -1.0)
    u = torch.full_like(x, 1.0)
    nbd = torch.full((n_loc,), 2, dtype=torch.int32, device=dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    t_setulb = 0.0
    iter_marks = []          # (wall, t_setulb) at each NEW_X

    def advance(iters):
        nonlocal t_setulb
        done = 0
        while done < iters:
            t0 = time.perf_counter()
            task = sol.setulb(x, l, u, nbd, g, 0.0, 0.0)
            t_setulb += time.perf_counter() - t0
            if task.startswith("FG"):
                sol.objective(0, x, g, deferred=True)   # f rides back with the next call's sums
            elif task.startswith("NEW_X"):
                done += 1
                iter_marks.append((time.perf_counter(), t_setulb))
            else:
                raise SystemExit("solver stopped: " + task)

    barrier()
    tw0 = time.perf_counter()
    advance(1)
    barrier()
    first_iter_s = time.perf_counter() - tw0
    nseg_first = int(sol.isave[32])
    # Untimed until the memory is full (col == m): whatever --warmup says, at least m + 1
    # iterations run first, so that every timed launch streams all 2m columns of W and the
    # roofline bytes below (computed for col = m) are the bytes each timed launch really moved.
    warm_done = 1
    warm_min = max(a.warmup, m + 1)
    while warm_done < warm_min or int(sol.isave[27]) < m:
        advance(1)
        warm_done += 1
        if warm_done > warm_min + 4 * m + 20:
            raise SystemExit("col never reached m = %d (skipped updates?)" % m)
    barrier()
    ts0 = t_setulb
    st0 = sol.stats()
    sol.pass_clock(1)            # hipEvents around every launch of the three W passes
    cols_timed = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        advance(1)
        cols_timed.append(int(sol.isave[27]))
    barrier()
    dt = time.perf_counter() - t0
    clocks = sol.pass_clock(0)   # {pass: (ms_total, launches)} over the timed region
    assert min(cols_timed) == max(cols_timed) == m, cols_timed
    dt_setulb = t_setulb - ts0
    if world > 1:
        tt = torch.tensor([dt, dt_setulb], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, dt_setulb = float(tt[0]), float(tt[1])
    stats = sol.stats()
    f_final = float(sol.f[0])
    col = int(sol.isave[27])
    nfree = int(sol.isave[37])

    # ---- roofline, live, hipEvents on the solver's stream ----
    # (1) the kernel that carries the WS/WY matvec INSIDE the iteration: cmprlb_wtv_kernel
    #     (r of cmprlb + W'r of subsm + formk's new row sums in one pass); algorithmic bytes per
    #     row = 2col reads of W + x, g reads (fp64) + iwhere (1 byte); xcp and r stay in registers
    #     (subsm_update_kernel recomputes it), so the pass writes nothing but its partials
    # (2) the bare W'v kernel (wtv_kernel), (2col+1) n s bytes -- BASELINE.md's definition
    head = int(sol.isave[26])
    mc = 5 if col <= 5 else 10 if col <= 10 else 20 if col <= 20 else 32
    # load policy the library picked (solver.hip init): nontemporal when W >> Infinity Cache
    ld_rows = (n_loc + 31) // 32 * 32
    nt = 2 * ld_rows * m * rbytes > (192 << 20)
    if os.environ.get("LBFGSB_NT") in ("0", "1"):
        nt = os.environ["LBFGSB_NT"] == "1"
    nts = "true" if nt else "false"

    def traffic_of(name, rows):
        tf = os.path.join(ROOT, "profiles", name)
        if os.path.exists(tf) and not a.real32 and col == 10:   # measured for fp64, col = 10
            try:
                return json.load(open(tf)).get("hbm_bytes_per_row") * rows
            except Exception:
                return None
        return None

    def in_run(name):
        ms, cnt = clocks[name]
        return (ms / cnt, cnt) if cnt else (None, 0)

    # Average launch duration over the timed region (in-run hipEvents); the same kernel launched
    # back to back after the run is kept beside it as a cross-check.  Passes over W of one
    # iteration: update_scan_kernel (read-only: the WS/WY matvecs W'd of cauchy and S'y, S's of
    # matupd, + formk's new row), subsm_update_kernel (W wv; the one pass that stores) and -- only
    # where the closed form for W'Z r does not apply (m > 10, few free variables, long walks) --
    # cmprlb_wtv_kernel (r of cmprlb + W'r of subsm).
    tname = "float" if a.real32 else "double"

    def pass_traffic(key, rows):
        tf = os.path.join(ROOT, "profiles", "w_pass_traffic.json")
        if os.path.exists(tf) and not a.real32 and col == 10:
            try:
                return json.load(open(tf))[key]["hbm_bytes_per_row"] * rows
            except Exception:
                return None
        return None

    def pass_record(key, which, label, alg_bytes, stores, traffic_key):
        ms_iso = sol.kernel_time(which, x, g, col, head, a.roofline_reps)
        ms_run, cnt = in_run(key)
        ms = ms_run if ms_run is not None else ms_iso
        ach = alg_bytes / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": label, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": pass_traffic(traffic_key, n_loc),
                "traffic_source": "profiles/w_pass_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                  "passes over this bench at n=1e8, in-iteration launches of this kernel "
                                  "[profiles/r03d_*]: bytes per row x rows; not re-measured in this run)",
                "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": ms,
                "launches_timed": cnt,
                "timing": ("hipEvents around each launch inside the timed region" if cnt else
                           "not launched inside the timed region: back-to-back launches after it"),
                "avg_launch_ms_back_to_back": ms_iso, "rows_per_launch": n_loc, "col": col,
                "stores": stores}
    n_sub = in_run("subsm_update")[1]
    rec_us = pass_record("update_scan", 4, "update_scan_kernel<%s, %d, %s> (run as the evaluation of the "
                         "first trial point)" % (tname, mc, nts),
                         ((2 * (col - 1) + 6) * rbytes + 2) * n_loc, "none", "update_scan")
    # the subspace pass stores t, r, the first trial point x and the pending Ws/Wy column; z and
    # d = x - t stay implicit while the unit first trial step stands (LBFGSB_LEAN=0: stored too)
    lean = os.environ.get("LBFGSB_LEAN", "1") != "0"
    n_st = 5 if lean else 6 + (1 if n_sub else 0)
    rec_su = pass_record("subsm_update", 3, "subsm_update_kernel<%s, %d, %s> (pending pair committed)"
                         % (tname, mc, nts), ((2 * col + 4 + n_st) * rbytes + 2) * n_loc,
                         ("t, r, trial x + Ws/Wy column (5 of %d streams)" % (2 * col + 9)) if lean else
                         ("z, d, t, r, trial x + Ws/Wy column (7 of %d streams)" % (2 * col + 11)),
                         "subsm_update")
    rec_cw = pass_record("cmprlb_wtv", 2, "cmprlb_wtv_kernel<%s, %d, true, %s>" % (tname, mc, nts),
                         ((2 * col + 2) * rbytes + 1) * n_loc, "none", "cmprlb_wtv")
    closed_steps, three_steps, handed_windows = sol.path_counts()
    if rec_cw["launches_timed"]:      # three-pass iteration: cmprlb_wtv carries W'r inside it
        roofline, others = rec_cw, [rec_us, rec_su]
    else:                             # two-pass iteration: the update pass carries the matvecs
        rec_cw["note"] = "not part of this run's iteration (W'Z r in closed form); timed back to back"
        roofline, others = rec_us, [rec_su, rec_cw]
    ms_kernel = sol.wtv_time(g, col, head, a.roofline_reps)
    alg_bytes = (2 * col + 1) * n_loc * rbytes
    achieved = alg_bytes / (ms_kernel * 1e-3) / 1e9
    roofline_wtv = {"bound": "hbm", "kernel": "wtv_kernel<%s, %d, %s>" % ("float" if a.real32 else "double", mc, nts),
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_of("wtv_traffic.json", n_loc),
                    "traffic_source": "profiles/wtv_traffic.json (static PMC measurement, see roofline)",
                    "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": ms_kernel,
                    "rows_per_launch": n_loc, "col": col}

    out = {
        "metric": "setulb iters/sec + achieved HBM GB/s on WS/WY matvec, n=1e8 m=10",
        "value": a.steps / dt,
        "unit": "iters/sec",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "warmup_run": warm_done,
        "col_min_timed": min(cols_timed),
        "col_max_timed": max(cols_timed),
        "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32" if a.real32 else "f64",
        "data": "synthetic",
        "config": {"workload": "separable bounded quadratic (SURVEY.md 8d), n=%d, m=%d, %s, "
                               "l=-1,u=1,x0=0, on-device objective"
                               % (n, m, "fp32 storage/fp64 accumulate" if a.real32 else "fp64"),
                   "n": n, "m": m, "rows_per_gpu": n_loc, "parallelism": "rows/%d" % world,
                   "collective": collective},
        "iters_per_sec_setulb_only": a.steps / dt_setulb,
        "first_iteration_s": first_iter_s,
        "first_iteration_nseg": nseg_first,
        "f_final": f_final,
        "col": col,
        "nfree": nfree,
        "host_syncs_per_iter": (stats["syncs"] - st0["syncs"]) / a.steps,
        "kernel_launches_per_iter": (stats["launches"] - st0["launches"]) / a.steps,
        "host_blocked_ms_per_iter": (stats["wait_seconds"] - st0["wait_seconds"]) / a.steps * 1e3,
        "cauchy_fullsorts": stats["cauchy_fullsorts"],
        "subspace_steps_closed_form": closed_steps,
        "subspace_steps_three_pass": three_steps,
        "cauchy_walks_served_by_update_pass": handed_windows,
        "roofline": roofline,
        "roofline_wtv": roofline_wtv,
        "roofline_other_w_passes": others,
    }
    sol.close()
    # the opt-in closed-form GCP (LBFGSB_F_PARALLEL_GCP): only the first iteration differs
    # (col = 0, nseg ~ 0.977 n); timed on a fresh context, outside the timed region above
    try:
        if world > 1:      # single-GPU leg only: no second communicator inside the scaling runs
            raise RuntimeError("skipped for n_gpus > 1")
        sol2 = lbfgsb_amd.DeviceSolver(n_loc, m, n_global=n, row0=row0, device=local_rank,
                                       same_stream_objective=True, real32=a.real32,
                                       parallel_gcp=True)
        x.zero_()
        barrier()
        tp0 = time.perf_counter()
        while True:
            task = sol2.setulb(x, l, u, nbd, g, 0.0, 0.0)
            if task.startswith("FG"):
                sol2.f[0] = sol2.objective(0, x, g)
            else:
                break
        barrier()
        out["first_iteration_parallel_gcp_s"] = time.perf_counter() - tp0
        out["first_iteration_parallel_gcp_nseg"] = int(sol2.isave[32])
        sol2.close()
    except Exception as e:
        out["first_iteration_parallel_gcp_s"] = None
        out["first_iteration_parallel_gcp_error"] = repr(e)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(m, n, min(a.cpu_n, n))
        except Exception as e:  # the baseline must never take the bench line down
            out["cpu_baseline"] = {"value": None, "unit": "iters/sec", "cores": 1, "kind": "reference",
                                   "sample": "failed: %r" % (e,)}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

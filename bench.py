#!/usr/bin/env python3
"""bench.py -- setulb iterations/sec on the MI355X-native L-BFGS-B inner iteration.

Workload (BASELINE.json metric): separable bounded quadratic, n = 1e8, m = 10, fp64,
l = -1, u = +1, x0 = 0, factr = pgtol = 0, on-device objective.  A "step"
is one L-BFGS-B iteration (one NEW_X return): every kernel of the hot path runs, plus the
f/g evaluations the line search asks for.  With --gpus N the n rows are sharded over N
ranks (STRONG scaling: n stays 1e8) and every reduction is completed with ONE RCCL collective
per host sync (all-gather of the <= 8m+15 partials, reduced in rank order on every rank).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n ROWS] [--m M]

Prints ONE JSON line (rank 0).  `value` = K / wall time of the K timed iterations
(objective evaluations included; `iters_per_sec_setulb_only` excludes them).
`roofline` is the DOMINANT kernel of the iteration (the one storing pass, subsm_update_kernel),
timed live with HIP events on the solver's stream inside the timed region, its `traffic` counted in
this run by two `rocprofv3 --pmc` child runs of this file after the timed legs (live_traffic()); the read-only pass
that carries the WS/WY matvecs (update_scan_kernel) and the bare W'v kernel are reported beside
it.  `cpu_baseline` times the real reference (oracle/_ref) on one host core, rank 0 at N=1 only,
on a bounded sample, and quotes the one full-size run on file.  `other_configs` (N=1 only): short
legs for BASELINE.json configs[1], [2], [4] and the per-rank shape of configs[3].
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
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short legs for the other BASELINE.json configs (N=1 only anyway)")
    ap.add_argument("--cpu-n", type=int, default=20_000_000,
                    help="rows of the CPU reference sample (same problem, same m)")
    ap.add_argument("--rccl-self", action="store_true",
                    help="single GPU: attach a 1-rank RCCL communicator, so that every reduction pays a "
                         "real RCCL collective + D2H + sync (latency floor of the sharded path; use with "
                         "--rows 12500000 = the per-rank shape of n=1e8 over 8 GPUs)")
    ap.add_argument("--allow-gloo-fallback", action="store_true",
                    help="multi-GPU: if the RCCL communicator cannot be created, complete the reductions "
                         "through a gloo host group instead of exiting non-zero")
    ap.add_argument("--classic", action="store_true",
                    help="drive lbfgsb_hip_setulb_dev (t = x, r = g as copies: 5 store streams in the storing "
                         "pass) instead of the ping-pong entry lbfgsb_hip_setulb_dev_pp (3 store streams)")
    ap.add_argument("--rosenbrock", action="store_true",
                    help="profiling aid: the headline leg runs the extended Rosenbrock objective with the drivers' "
                         "box (BASELINE.json configs[2] with --n 10000000) instead of the separable quadratic; "
                         "the metric line is then labelled accordingly")
    ap.add_argument("--no-defer", action="store_true",
                    help="contexts WITHOUT LBFGSB_F_DEFER_LNSRCH: every FG_LNSRCH return waits for the storing "
                         "pass's sums (one more host sync per iteration; what an ordinary caller gets)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="lbfgsb_hip_set_option on every context of the run (A/B measurements)")
    ap.add_argument("--no-compact", action="store_true",
                    help="contexts WITHOUT the option compact_w: the two passes over W stream every row of every "
                         "column under the iwhere mask (what every round before 6 measured) instead of running on the "
                         "tile-local free-row layout (fp64, m <= 10: DESIGN.md 4g)")
    ap.add_argument("--roofline-reps", type=int, default=20)
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="skip the two rocprofv3 --pmc child runs that count the HBM bytes of the passes over W "
                         "(N=1 only; skipped by itself when this process already runs under a profiler)")
    ap.add_argument("--pmc-child", action="store_true",
                    help="(internal) the run rocprofv3 counts: warm-up + timed iterations, then exit")
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


def under_profiler():
    pre = os.environ.get("LD_PRELOAD", "")
    return "rocprof" in pre or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)


def live_traffic(a, n_loc, timeout_s=240):
    """HBM bytes per launch of the passes over W, COUNTED IN THIS RUN: two child runs of this file under
    `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `rocprofv3 --pmc WRITE_SIZE --kernel-trace` (separate
    passes, no other trace domain, the program itself after `--`: MI355X_MICROARCH.md, HBM / rocprofv3).
    The child (--pmc-child) runs the same workload through the same entry: START, warm-up until the memory
    is full, a few more iterations, exit.  Corrections for gfx950 as that guide prescribes: both counters
    are in KiB, FETCH_SIZE reports half of the bytes of a wide coalesced streaming read (x2), WRITE_SIZE is
    exact.  Launches at col = m are the ones whose counter is within 1 % of the largest seen for that kernel
    (earlier launches read fewer columns).  Returns {kernel family: bytes per launch} or raises."""
    import csv
    import glob
    import shutil
    import signal
    import statistics
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    base = tempfile.mkdtemp(prefix="lbfgsb_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--steps", "4", "--warmup",
             str(a.warmup), "--n", str(a.n), "--m", str(a.m), "--no-cpu-baseline", "--no-other-configs",
             "--no-live-traffic"]
    child += ["--real32"] if a.real32 else []
    child += ["--classic"] if a.classic else []
    child += ["--no-compact"] if a.no_compact else []
    for kv in a.opt:
        child += ["--opt", kv]
    counted = {}
    t0 = time.perf_counter()
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(base, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--"] + child
            left = timeout_s - (time.perf_counter() - t0)
            if left < 20:
                raise RuntimeError("no time left for the %s pass" % counter)
            # own process group: on a timeout the profiler AND the program it started are ended (by the
            # exact group id started here)
            pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                                  start_new_session=True)
            try:
                _, err = pr.communicate(timeout=left)
            except subprocess.TimeoutExpired:
                os.killpg(pr.pid, signal.SIGKILL)
                pr.communicate()
                raise RuntimeError("the %s pass did not finish in %.0f s" % (counter, left))
            if pr.returncode != 0:
                raise RuntimeError("rocprofv3 --pmc %s exited %d: %s" % (counter, pr.returncode,
                                                                        err.decode(errors="replace")[-300:]))
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                raise RuntimeError("no counter_collection.csv from the %s pass" % counter)
            per = {}
            for f in files:
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if r["Counter_Name"] == counter:
                            per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
            counted[counter] = per
    finally:
        shutil.rmtree(base, ignore_errors=True)
    out = {}
    for fam in ("subsm_update_kernel", "update_scan_kernel"):
        # the instantiation of this family that moved the most bytes = the one the timed iterations launch
        fk = max((k for k in counted["FETCH_SIZE"] if fam in k), key=lambda k: sum(counted["FETCH_SIZE"][k]),
                 default=None)
        if fk is None or fk not in counted["WRITE_SIZE"]:
            continue
        f, w = counted["FETCH_SIZE"][fk], counted["WRITE_SIZE"][fk]
        f_in = [v for v in f if v >= 0.99 * max(f)]
        w_in = [v for v in w if v >= 0.99 * max(w)] if max(w) > 0.01 * max(f) else w
        rd, wr = 2.0 * 1024.0 * statistics.median(f_in), 1024.0 * statistics.median(w_in)
        out[fam] = {"kernel": fk[:120], "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                    "hbm_bytes_per_launch": rd + wr, "hbm_bytes_per_row": (rd + wr) / n_loc,
                    "launches_counted": len(f_in), "FETCH_SIZE_KiB_raw_median": statistics.median(f_in),
                    "WRITE_SIZE_KiB_raw_median": statistics.median(w_in)}
    out["seconds"] = time.perf_counter() - t0
    return out


def cpu_baseline(m, n_full, n_sample):
    """The untouched reference (oracle/_ref, amdflang -O2; the -fdefault-integer-8 build, which is
    the one that can run n = 1e8 at all) on ONE host core -- it is single
    threaded by construction -- on the same problem at n_sample rows.  Iterations with col = m
    are timed inside setulb only (the objective is excluded); the value is scaled linearly in n
    to n_full rows.  The one FULL-SIZE run on file (profiles/r3a_cpu_ref_full_n1e8_m10.json, the same
    build on a GPU box's host) is quoted beside it: at n = 1e8 the reference is slower than the linear
    scaling of the sample says (caches, TLB), `extrapolated_over_measured_full_size` says by how much."""
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
    out = {
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
    full_file = os.path.join(ROOT, "profiles", "r3a_cpu_ref_full_n1e8_m10.json")
    if os.path.exists(full_file) and n_full == 100_000_000 and m == 10:
        try:
            d = json.load(open(full_file))
            out["full_size_run_on_file"] = {
                "source": "profiles/r3a_cpu_ref_full_n1e8_m10.json (profiles/scripts/cpu_ref_full.py: the same "
                          "reference build, n = 1e8, m = 10, 16 iterations, one core; not re-run here)",
                "iters_per_sec": d["iters_per_sec_col_eq_m"], "s_per_iter": d["s_per_iter_col_eq_m"],
                "iters_timed": d["iters_timed_col_eq_m"], "first_iteration_s": d["first_iteration_s"],
                "peak_rss_gb": d["peak_rss_gb"], "host_cpu_model": d["host"]["model"],
            }
            out["extrapolated_over_measured_full_size"] = out["value"] / d["iters_per_sec_col_eq_m"]
            # `value` is ALWAYS this run's live bounded sample scaled linearly in n (never a number from a file);
            # the one full-size measurement on file stands beside it under its own name.  At n = 1e8 the
            # reference is slower than the linear scaling says (caches, TLB): the ratio above says by how much.
            out["value_source"] = "live bounded sample of this run, scaled linearly in n"
            out["value_full_size_on_file"] = d["iters_per_sec_col_eq_m"]
        except Exception:   # noqa: BLE001
            pass
    return out


def self_launch(a):
    """`python bench.py --gpus N` given bare (no launcher around it): start
    `python -m torch.distributed.run --nproc-per-node N bench.py <same args>` as a CHILD process -- before this
    process has made any GPU call, so nothing that has initialised the GPU is ever replaced -- forward rank 0's
    JSON line and exit with the child's code.  Under a launcher (WORLD_SIZE set) this is never reached."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    pr = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    printed = 0
    for line in pr.stdout:            # (stderr goes straight through)
        if line.startswith("{") and not printed:
            sys.stdout.write(line)
            sys.stdout.flush()
            printed += 1
        else:
            sys.stderr.write(line)
    rc = pr.wait()
    if rc == 0 and not printed:
        sys.stderr.write("bench.py: the launched ranks exited 0 without a JSON line\n")
        rc = 1
    return rc


GOLDEN = os.path.join(ROOT, "tests", "golden")
# (n, m, objective kind) -> rows the REAL reference printed for that workload at full size, and the tolerance on f
# the GPU parity tests use for the same fixture (tests/test_gpu_parity.py)
GOLDEN_ROWS = {(100_000_000, 10, 0): ("quad_n1e8_m10_ref_rows.json", 1e-9),
               (1_000_000, 10, 0): ("quad_n1e6_m10_ref_rows.json", 1e-9),
               (20_000_000, 48, 0): ("quad_n2e7_m48_ref_rows.json", 1e-9),
               (50_000_000, 32, 0): ("quad_n5e7_m32_ref_rows.json", 1e-9),
               (10_000_000, 10, 1): ("rosenbrock_n1e7_anchors.json", 1e-7)}
# REAL32 contexts against the REAL64 reference's rows of the same shape: tests/test_gpu_real32.py's rules (nfg exactly,
# nseg / nfree within 1e-3 relative + 5, f to 1e-6)
GOLDEN_ROWS_R32 = {(100_000_000, 20, 0): ("quad_n1e8_m20_ref_rows.json", 1e-6)}


def parity_in_run(rows, n, m, real32, kind):
    """The timed path checks itself: the (iter, nfg, nseg, nfree, f) of every NEW_X return of a leg -- the very
    entry, flags and buffers that are timed -- against the rows the REAL reference (oracle/_ref) printed for this
    workload at full size (tests/golden/*.json: data; generating scripts profiles/scripts/cpu_ref_full.py,
    tests/golden/make_anchors_n1e7.py).  Integers exactly, f within the fixture's tolerance.  Shapes without
    such rows report rows_checked = 0."""
    fx = (GOLDEN_ROWS_R32 if real32 else GOLDEN_ROWS).get((n, m, kind))
    if fx is None or not os.path.exists(os.path.join(GOLDEN, fx[0])):
        return {"rows_checked": 0, "ok": None, "why": "no reference rows on file for this shape"}
    by_iter = {r["iter"]: r for r in json.load(open(os.path.join(GOLDEN, fx[0])))["rows"]}
    checked, bad, ok_before = 0, [], 0
    for it, nfg, nseg, nfree, f in rows:
        w = by_iter.get(it)
        if w is None:
            continue
        checked += 1
        if real32:
            ok = (nfg == w["nfg"] and abs(nseg - w["nseg"]) <= 1e-3 * w["nseg"] + 5 and
                  abs(nfree - w["nfree"]) <= 1e-3 * w["nfree"] + 5 and abs(f - w["f"]) <= fx[1] * abs(w["f"]))
        else:
            ok = (nfg, nseg, nfree) == (w["nfg"], w["nseg"], w["nfree"]) and abs(f - w["f"]) <= fx[1] * abs(w["f"])
        if ok and not bad:
            ok_before += 1
        if not ok and len(bad) < 3:
            bad.append({"got": [it, nfg, nseg, nfree, f], "want": [w["iter"], w["nfg"], w["nseg"], w["nfree"], w["f"]]})
    out = {"rows_checked": checked, "ok": checked > 0 and not bad, "rows_ok_before_first_split": ok_before,
           "f_rel_tol": fx[1],
           "iters": [rows[0][0], rows[-1][0]] if rows else [], "against": "tests/golden/" + fx[0]}
    if bad:
        out["mismatch"] = bad
    return out


def problem_tensors(torch, dev, kind, n_loc, row0, real32, arbitrary_box=False):
    """x0, l, u, nbd of this rank's rows: kind 0 = separable bounded quadratic,
    kind 1 = extended Rosenbrock with the drivers' box (test/driver1.f90:233-251).
    arbitrary_box (kind 0): every l_i, u_i its own value (the box [-1, 1] widened by up to 1e-3 per side, a
    closed-form function of the row number) -- neither the uniform-bounds constants nor the dictionary apply,
    the passes stream l, u and nbd: what a caller with per-variable bounds gets."""
    rdt = torch.float32 if real32 else torch.float64
    if kind == 0:
        x = torch.zeros(n_loc, dtype=rdt, device=dev)
        l = torch.full_like(x, -1.0)
        u = torch.full_like(x, 1.0)
        if arbitrary_box:
            i = torch.arange(row0 + 1, row0 + n_loc + 1, dtype=torch.int64, device=dev)
            l -= ((7919 * i) % 1000003).to(rdt) * 1.0e-9
            u += ((104729 * i) % 1000033).to(rdt) * 1.0e-9
            del i
    else:
        x = torch.full((n_loc,), 3.0, dtype=rdt, device=dev)
        u = torch.full_like(x, 100.0)
        l = torch.full_like(x, -100.0)
        l[(row0 % 2)::2] = 1.0          # odd 1-based global index
    nbd = torch.full((n_loc,), 2, dtype=torch.int32, device=dev)
    return x, l, u, nbd


class Run:
    """one solver context + its problem, advanced iteration by iteration"""

    def __init__(self, torch, dist, la, a, *, n, m, real32, kind, world, rank, local_rank, rccl_self, opts,
                 parallel_gcp=False, defer=None, classic=None, arbitrary_box=False, torch_objective=False):
        self.torch, self.dist, self.world = torch, dist, world
        self.kind, self.n, self.m = kind, n, m
        dev = torch.device("cuda", local_rank)
        self.dev = dev
        row0, n_loc = la.block_partition(n, world, rank)
        self.n_loc = n_loc
        # torch_objective: f, g by torch ops on torch's current stream, ordered against the solver's stream with events
        # in both directions (stream_ordered: lbfgsb_hip_return_event / lbfgsb_hip_wait_stream), f handed over as a
        # device scalar (lbfgsb_hip_f_device) -- what an ordinary PyTorch caller does, with no host sync of its own
        self.torch_objective = bool(torch_objective)
        mk = lambda: la.DeviceSolver(n_loc, m, n_global=n, row0=row0, device=local_rank,   # noqa: E731
                                     same_stream_objective=not torch_objective, stream_ordered=torch_objective,
                                     real32=real32, options=opts,
                                     parallel_gcp=parallel_gcp,
                                     defer_lnsrch=(not a.no_defer) if defer is None else bool(defer))
        self.sol = mk()
        self.collective = "none"
        if world > 1:
            # RCCL on the solver's stream.  A scaling curve must never silently be a gloo curve: if
            # the communicator cannot be created on some rank the run exits non-zero, unless
            # --allow-gloo-fallback asks for the (tiny) reductions to go through a gloo host group.
            ok = 1
            with stdout_to_stderr():
                try:
                    la.attach_rccl(self.sol, rank, world, dev)
                except Exception as e:   # noqa: BLE001
                    ok = 0
                    sys.stderr.write("rank %d: RCCL communicator failed (%r)\n" % (rank, e))
                flag = torch.tensor([ok], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                self.collective = ("one RCCL all-gather of the <=8m+15 fp64 partials per host sync, reduced in "
                                   "rank order on every rank")
            elif not a.allow_gloo_fallback:
                self.sol.close()
                dist.destroy_process_group()
                raise SystemExit("RCCL communicator unavailable on some rank (pass --allow-gloo-fallback "
                                 "to measure with a gloo host group instead)")
            else:
                self.sol.close()
                self.sol = mk()
                la.attach_host_group(self.sol, rank, world, group=dist.new_group(backend="gloo"))
                self.collective = "gloo host all-reduce (RCCL communicator unavailable; --allow-gloo-fallback)"
        elif rccl_self:
            with stdout_to_stderr():
                la.attach_rccl(self.sol, 0, 1, dev)
            self.collective = "RCCL all-gather on a 1-rank communicator (--rccl-self: latency floor)"
        x, self.l, self.u, self.nbd = problem_tensors(torch, dev, kind, n_loc, row0, real32, arbitrary_box)
        if torch_objective:
            assert kind == 0 and not real32
            i = torch.arange(row0 + 1, row0 + n_loc + 1, dtype=torch.int64, device=dev)
            self.a_ = 1.0 + 99.0 * ((7919 * i) % 10007).to(torch.float64) / 10006.0
            self.c_ = -2.0 + 4.0 * ((104729 * i) % 100003).to(torch.float64) / 100002.0
            del i
            self.tmp_ = torch.empty_like(x)
        self.pp = not (a.classic if classic is None else classic)
        # ping-pong entry: two pairs of iterate / gradient buffers, the library tells which one is live
        self.xs = [x, torch.empty_like(x)] if self.pp else [x]
        self.gs = [torch.zeros_like(x), torch.empty_like(x)] if self.pp else [torch.zeros_like(x)]
        self.cur = 0
        self.t_setulb = 0.0
        self.rows = []        # (iter, nfg, nseg, nfree, f) at every NEW_X return of this context
        self.stamps = []      # host clock at those returns (time-to-solution; no sync: what the caller sees)
        self.t_start = None   # host clock just before the START call

    @property
    def x(self):
        return self.xs[self.cur]

    @property
    def g(self):
        return self.gs[self.cur]

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def advance(self, iters):
        sol = self.sol
        done = 0
        while done < iters:
            t0 = time.perf_counter()
            if self.t_start is None:
                self.t_start = t0
            if self.pp:
                task, self.cur = sol.setulb_pp(self.xs, self.l, self.u, self.nbd, self.gs, 0.0, 0.0)
            else:
                task = sol.setulb(self.xs[0], self.l, self.u, self.nbd, self.gs[0], 0.0, 0.0)
            self.t_setulb += time.perf_counter() - t0
            if task.startswith("FG") and self.torch_objective:
                torch = self.torch
                d = torch.sub(self.x, self.c_, out=self.tmp_)
                torch.mul(self.a_, d, out=self.g)
                f_dev = 0.5 * torch.dot(self.g, d)
                sol.set_f_device(f_dev)          # (no .item(): this rank's part of f rides back with the next call's sums)
            elif task.startswith("FG"):
                sol.objective(self.kind, self.x, self.g, deferred=True)   # f rides back with the next call's sums
            elif task.startswith("NEW_X"):
                done += 1
                isv = sol.isave
                self.rows.append((int(isv[29]), int(isv[33]), int(isv[32]), int(isv[37]), float(sol.f[0])))
                self.stamps.append(time.perf_counter())
            else:
                raise RuntimeError("solver stopped: " + task)

    def close(self):
        import gc
        self.sol.close()
        self.xs = self.gs = self.l = self.u = self.nbd = None
        self.a_ = self.c_ = self.tmp_ = None
        # (release the 10+ GB of this leg NOW: a cyclic-garbage pass that frees them in the middle of the next
        #  leg's timed region costs that leg a 30-70 ms device-wide stall)
        gc.collect()
        self.torch.cuda.empty_cache()


def timed_leg(run, steps, warm_min, need_full_memory=True):
    """first iteration apart, then untimed until the memory is full (col == m) and at least
    warm_min iterations have run, then `steps` timed iterations bracketed by barriers."""
    sol, m = run.sol, run.m
    run.barrier()
    tw0 = time.perf_counter()
    run.advance(1)
    run.barrier()
    first_iter_s = time.perf_counter() - tw0
    nseg_first = int(sol.isave[32])
    warm_done = 1
    while warm_done < warm_min or (need_full_memory and int(sol.isave[27]) < m):
        run.advance(1)
        warm_done += 1
        if warm_done > warm_min + 4 * m + 20:
            raise RuntimeError("col never reached m = %d (skipped updates?)" % m)
    # option compact_w: the layout is packed in the first iteration with the memory full and a settled free set (a
    # one-off re-sort of all of W): that iteration belongs to the warm-up, not to the timed region
    extra = 0
    while extra < 6:
        packs, _unpacks, packed, eligible = sol.compact_stats()
        nf = int(sol.isave[37])
        if not eligible or packed or nf > 0.9 * run.n or packs > 0:
            break
        run.advance(1)
        warm_done += 1
        extra += 1
    run.barrier()
    ts0, st0 = run.t_setulb, sol.stats()
    hg0 = sol.host_gap()
    hs0 = sol.host_segments()
    sol.pass_clock(1)            # hipEvents around every launch of the three W passes
    cols_timed, step_ms = [], []
    t0 = time.perf_counter()
    tp = t0
    for _ in range(steps):
        run.advance(1)
        cols_timed.append(int(sol.isave[27]))
        tn = time.perf_counter()      # (host clock at the NEW_X return: no extra sync, the value is K / dt)
        step_ms.append((tn - tp) * 1e3)
        tp = tn
    run.barrier()
    dt = time.perf_counter() - t0
    clocks = sol.pass_clock(0)   # {pass: (ms_total, launches)} over the timed region
    dt_setulb = run.t_setulb - ts0
    dt_min = dt
    if run.world > 1:
        tt = run.torch.tensor([dt, dt_setulb, -dt], dtype=run.torch.float64, device=run.dev)
        run.dist.all_reduce(tt, op=run.dist.ReduceOp.MAX)
        dt, dt_setulb, dt_min = float(tt[0]), float(tt[1]), -float(tt[2])
    st1 = sol.stats()
    hg1 = sol.host_gap()
    st1["host_gap_us"] = (hg1[0] - hg0[0]) / max(1, hg1[1] - hg0[1]) * 1e6
    hs1 = sol.host_segments()
    names = ("linesearch_and_return", "caller_new_x_to_reentry", "termination_matupd_formt", "cauchy_freev",
             "formk_closed_form_solves")
    st1["host_segments_us"] = {nm: (b - a_) / max(1, hg1[1] - hg0[1]) * 1e6 for nm, a_, b in zip(names, hs0, hs1)}
    return dict(first_iter_s=first_iter_s, nseg_first=nseg_first, warm_done=warm_done, cols_timed=cols_timed,
                dt=dt, dt_min=dt_min, dt_setulb=dt_setulb, clocks=clocks, st0=st0, st1=st1,
                step_ms_median=float(np.median(step_ms)), step_ms_max=float(np.max(step_ms)))


def pass_bytes(col, rbytes, pp, lean=True, ub=0, wfrac=1.0):
    """algorithmic bytes per row of the two passes over W of an iteration;
    ub = lbfgsb_hip_uniform_bounds mask: bound arrays that hold one value are not streamed;
    wfrac = the fraction of the rows whose W entries a pass needs: 1 for the kernels that stream every row under
    a mask, nfree / n on the tile-local free-row layout (option compact_w) -- SURVEY.md 8(d)'s R = nf for the
    stored columns (src/lbfgsb.f90:1565-1583, :2743-2778, :1756-1793 run over Index(1:nfree)); the n-vectors, the
    pending pair's two vectors and the stores are per row either way"""
    bounds = (0 if ub & 1 else rbytes) + (0 if ub & 2 else rbytes) + (0 if ub & 4 else 1)   # l, u, nbd
    upd = 2 * (col - 1) * rbytes * wfrac + 4 * rbytes + 1 + bounds   # 2(col-1) W columns + x, g, r, t + iwhere + bounds
    n_st = (3 if pp else 5) if lean else (5 if pp else 7)
    # 2 (col - 1) stored W columns + the pending pair from r, t + x, g + iwhere + bounds + stores
    sub = 2 * (col - 1) * rbytes * wfrac + (4 + n_st) * rbytes + 1 + bounds
    return upd, sub, n_st


def other_config(torch, dist, la, a, name, *, n, m, real32, kind, rccl_self, steps, warm_min, local_rank, opts,
                 defer=None, classic=None, arbitrary_box=False, torch_objective=False):
    """a short leg for one of the other BASELINE.json configs: it/s and the two pass fractions"""
    run = Run(torch, dist, la, a, n=n, m=m, real32=real32, kind=kind, world=1, rank=0, local_rank=local_rank,
              rccl_self=rccl_self, opts=opts, defer=defer, classic=classic, arbitrary_box=arbitrary_box,
              torch_objective=torch_objective)
    try:
        coll_us = run.sol.collective_time(1000) if rccl_self else None
        r = timed_leg(run, steps, warm_min, need_full_memory=(kind == 0))
        col = int(run.sol.isave[27])
        rb = 4 if real32 else 8
        ub = run.sol.uniform_bounds()
        cst = run.sol.compact_stats()
        wfrac = (int(run.sol.isave[37]) / float(n)) if (cst[3] and cst[2]) else 1.0   # rows whose W entries the passes need
        upd_b, sub_b, _ = pass_bytes(max(col, 1), rb, run.pp, opts.get("lean", 1) != 0, ub, wfrac)
        passes = {}
        cw_b = (2 * max(col, 1) + 2) * rb + 1     # the third pass of col > 20: 2 col W columns + x, g + iwhere
        for key, bpr in (("update_scan", upd_b), ("subsm_update", sub_b), ("cmprlb_wtv", cw_b)):
            ms, cnt = r["clocks"][key]
            if cnt:
                ach = bpr * run.n_loc / (ms / cnt * 1e-3) / 1e9
                passes[key] = {"avg_launch_ms": ms / cnt, "launches_timed": cnt, "achieved_GBs": ach,
                               "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_row": bpr}
        closed, three, _ = run.sol.path_counts()
        return {"config": name, "n": n, "m": m, "dtype": "f32" if real32 else "f64", "value": steps / r["dt"],
                "unit": "iters/sec", "ms_per_step": r["dt"] / steps * 1e3, "steps": steps,
                "ms_per_step_median": r["step_ms_median"], "ms_per_step_max": r["step_ms_max"],
                "warmup_run": r["warm_done"], "col_timed": [min(r["cols_timed"]), max(r["cols_timed"])],
                "first_iteration_s": r["first_iter_s"], "first_iteration_nseg": r["nseg_first"],
                "host_syncs_per_iter": (r["st1"]["syncs"] - r["st0"]["syncs"]) / steps,
                "freev_passes_skipped_per_iter": (r["st1"]["freev_skipped"] - r["st0"]["freev_skipped"]) / steps,
                "passes": passes, "subspace_steps_closed_form": closed, "subspace_steps_three_pass": three,
                "tie_splits": run.sol.tie_splits(), "collective": run.collective, "f_final": float(run.sol.f[0]),
                "collective_us": coll_us[0] if coll_us else None, "collective_us_min": coll_us[1] if coll_us else None,
                "lnsrch_setups_deferred_reissued": list(run.sol.defer_stats()),
                "host_algebra_us_between_passes": r["st1"]["host_gap_us"],
                "host_segments_us": r["st1"]["host_segments_us"],
                "uniform_bounds_mask": ub, "options": opts,
                "compact_w": {"eligible": cst[3], "packs": cst[0], "unpacks": cst[1], "packed": cst[2],
                              "w_rows_fraction": wfrac},
                # (the arbitrary-box leg is another problem than the one the reference's rows belong to)
                "parity_in_run": ({"rows_checked": 0, "ok": None, "why": "per-variable bounds: not the fixture's problem"}
                                  if arbitrary_box else parity_in_run(run.rows, n, m, real32, kind))}
    finally:
        run.close()


def host_form_leg(la, n=10_000_000, m=10, iters=26, warm=12):
    """The HOST-POINTER form, the reference's own argument list (what the Fortran module's setulb and
    lbfgsb_amd.setulb bind: lbfgsb_hip_setulb_host): x, g, wa live in host memory, the objective is evaluated on
    the host (numpy), and per iteration g travels H2D, the trial x and the previous-iterate slot of wa D2H --
    24 n bytes over PCIe.  Timed inside setulb only (the host objective is the caller's business) over the
    iterations warm+1 .. iters; the PCIe-inclusive rate is never the headline `value`.  Pinned by SURVEY.md 8c's
    anchors of this problem at n = 1e7: nseg(it1) = 9767199, nfree(it2) = 4999953."""
    i = np.arange(1, n + 1, dtype=np.int64)
    a_ = 1.0 + 99.0 * ((7919 * i) % 10007) / 10006.0
    c_ = -2.0 + 4.0 * ((104729 * i) % 100003) / 100002.0
    del i
    x, g = np.zeros(n), np.zeros(n)
    l, u = np.full(n, -1.0), np.full(n, 1.0)
    nbd = np.full(n, 2, np.int32)
    wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m)
    iwa = np.zeros(3 * n, np.int32)
    task, csave = la.solver.pad60("START"), la.solver.pad60("")
    lsave, isave, dsave, f = np.zeros(4, np.int32), np.zeros(44, np.int32), np.zeros(29), np.zeros(1)
    t_in = t_obj = 0.0
    marks, rows = {}, []
    t_first = None
    tmp = np.empty(n)
    try:
        while True:
            t0 = time.perf_counter()
            la.setulb(n, m, x, l, u, nbd, f, g, 0.0, 0.0, wa, iwa, task, -1, csave, lsave, isave, dsave)
            t_in += time.perf_counter() - t0
            ts = la.solver.task_str(task)
            if ts.startswith("FG"):
                t0 = time.perf_counter()
                np.subtract(x, c_, out=tmp)
                np.multiply(a_, tmp, out=g)
                f[0] = 0.5 * float(np.dot(g, tmp))
                t_obj += time.perf_counter() - t0
            elif ts.startswith("NEW_X"):
                it = int(isave[29])
                rows.append((it, int(isave[33]), int(isave[32]), int(isave[37])))
                marks[it] = (t_in, t_obj)
                if it == 1:
                    t_first = t_in
                if it >= iters:
                    break
            else:
                raise RuntimeError("host-form leg stopped: " + ts)
    finally:
        la.load_library().lbfgsb_hip_release_host(isave.ctypes.data)
    k = iters - warm
    dt_in, dt_obj = marks[iters][0] - marks[warm][0], marks[iters][1] - marks[warm][1]
    nfg = rows[-1][1] - rows[warm - 1][1]
    pcie = (nfg * 2 + k) * 8.0 * n        # g up + trial x down per evaluation, the t slot once per iteration
    anchors_ok = (n != 10_000_000 or m != 10) or (rows[0][2] == 9767199 and rows[1][3] == 4999953)
    return {"config": "host-pointer form (lbfgsb_hip_setulb_host: the reference's argument list, host arrays, host "
                      "objective), separable bounded quadratic n=%d, m=%d, fp64" % (n, m), "n": n, "m": m,
            "value": k / dt_in, "unit": "iters/sec inside setulb (PCIe transfers included, host objective excluded)",
            "ms_per_step": dt_in / k * 1e3, "steps": k, "iters_per_sec_with_host_objective": k / (dt_in + dt_obj),
            "host_objective_ms_per_eval": dt_obj / max(1, nfg) * 1e3, "pcie_bytes_per_iter": pcie / k,
            "pcie_GBs_inside_setulb": pcie / dt_in / 1e9, "first_iteration_s_inside_setulb": t_first,
            "anchors_ok": bool(anchors_ok), "rows_first2": rows[:2]}


LINE_CAP = 4096      # bytes: the driver keeps an 8 KB tail of stdout; the LAST stdout line must fit well inside


def _r(v, nd=6):
    """floats rounded to nd significant digits (the line is a record, not a data file)"""
    if isinstance(v, float):
        return float("%.*g" % (nd, v))
    if isinstance(v, dict):
        return {k: _r(x, nd) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, nd) for x in v]
    return v


def compact_line(out):
    """The ONE line the driver parses: the contract's keys + `roofline` + `cpu_baseline`, hard-capped at
    LINE_CAP bytes.  Everything else of `out` (other_configs, host_form, roofline_other_w_passes, the raw counters,
    prose) goes to bench_detail.json beside this file and to stderr."""
    c, rf, rw, cb = out["config"], out.get("roofline") or {}, out.get("roofline_wtv") or {}, out.get("cpu_baseline")
    par = c.get("parity_in_run") or {}
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {
        "workload": c["workload"][:160], "n": c["n"], "m": c["m"], "rows_per_gpu": c["rows_per_gpu"],
        "entry": c["entry"][:60], "defer_lnsrch": c["defer_lnsrch"], "uniform_bounds_mask": c["uniform_bounds_mask"],
        "compact_w": c.get("compact_w"),
        "parity_in_run": {k: par.get(k) for k in ("rows_checked", "ok", "iters", "f_rel_tol")},
        "rccl_nranks": c["rccl_nranks"], "first_iteration_s": c["first_iteration_s"],
        "time_to_30_iterations_s": c.get("time_to_30_iterations_s"),
    }
    if out["n_gpus"] > 1:   # what a scaling run needs to explain itself
        line["config"].update({
            "first_iteration_s_per_rank": c["first_iteration_s_per_rank"],
            "collective_us": out.get("collective_us"), "comm_kind": out.get("comm_kind"),
            "ms_per_step_rank_min": out["ms_per_step_rank_min"], "ms_per_step_rank_max": out["ms_per_step_rank_max"]})
    line["config"]["host_syncs_per_iter"] = out["host_syncs_per_iter"]
    line["config"]["collectives_per_iter"] = out["collectives_per_iter"]
    legs = c.get("legs")
    if legs:
        line["config"]["legs"] = legs
    line["roofline"] = {"bound": "hbm", "kernel": (rf.get("kernel") or "")[:100]}
    for k in ("achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "algorithmic_bytes_per_launch",
              "algorithmic_bytes_per_row", "avg_launch_ms", "launches_timed", "rows_per_launch"):
        line["roofline"][k] = rf.get(k)
    line["roofline_wtv"] = ({k: rw.get(k) for k in ("achieved", "frac", "avg_launch_ms", "algorithmic_bytes_per_launch")}
                            if rw else None)
    line["update_scan"] = {"frac": out.get("update_scan_frac_of_hbm_peak"), "avg_launch_ms": out.get("update_scan_ms_in_run")}
    if cb is not None:
        line["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "value_full_size_on_file",
                                                       "host_cpu_model", "n_sample", "s_per_iter_at_sample")}
        line["cpu_baseline"]["sample"] = (cb.get("sample") or "")[:200]
    line["detail"] = "bench_detail.json"
    line = _r(line)
    txt = json.dumps(line, separators=(",", ":"))
    if len(txt) > LINE_CAP and "legs" in line["config"]:      # (never expected: the legs are <= 1 KB)
        line["config"]["legs"] = {k: v[:40] for k, v in line["config"]["legs"].items()}
        txt = json.dumps(line, separators=(",", ":"))
    if len(txt) > LINE_CAP:
        line["config"].pop("legs", None)
        line["cpu_baseline"].pop("sample", None) if "cpu_baseline" in line else None
        txt = json.dumps(line, separators=(",", ":"))
    assert len(txt) <= LINE_CAP, len(txt)
    return txt


def emit(out):
    """detail -> bench_detail.json (+ stderr), then the compact line as the LAST line of stdout"""
    detail = json.dumps(out)
    try:
        with open(os.path.join(ROOT, "bench_detail.json"), "w") as fh:
            fh.write(detail + "\n")
    except OSError as e:
        sys.stderr.write("bench.py: bench_detail.json not written (%r)\n" % (e,))
    sys.stderr.write("bench.py detail: " + detail + "\n")
    sys.stderr.flush()
    print(compact_line(out), flush=True)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: be our own launcher (a child process; nothing here has touched the GPU)
        raise SystemExit(self_launch(a))
    import torch
    import torch.distributed as dist
    import lbfgsb_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (a.gpus, world))
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
    opts = {}
    for kv in a.opt:
        k, v = kv.split("=", 1)
        opts[k] = float(v)
    if not a.no_compact:
        # (contexts that cannot use it -- REAL32, m > 10 -- ignore the option: lbfgsb_hip_compact_stats says so)
        opts.setdefault("compact_w", 1.0)

    n, m = a.n, a.m
    rbytes = 4 if a.real32 else 8
    # the objective is the library's own kernel on the solver's stream, so the FG return needs no
    # host sync; contiguous block sharding of the rows
    run = Run(torch, dist, lbfgsb_amd, a, n=n, m=m, real32=a.real32, kind=1 if a.rosenbrock else 0, world=world,
              rank=rank, local_rank=local_rank, rccl_self=a.rccl_self, opts=opts)
    sol, n_loc = run.sol, run.n_loc
    # one host sync of the iteration by itself, before the run starts (every rank: it is a collective): what each of
    # the host_syncs_per_iter below costs on this communicator (all-gather + copy to mapped host memory + poll)
    coll_us = sol.collective_time(1000) if (world > 1 or a.rccl_self) else None
    # Untimed until the memory is full (col == m): whatever --warmup says, at least m + 1
    # iterations run first, so that every timed launch streams all 2m columns of W and the
    # roofline bytes below (computed for col = m) are the bytes each timed launch really moved.
    r = timed_leg(run, a.steps, max(a.warmup, m + 1), need_full_memory=not a.rosenbrock)
    cols_timed, clocks, dt, dt_setulb = r["cols_timed"], r["clocks"], r["dt"], r["dt_setulb"]
    assert a.rosenbrock or min(cols_timed) == max(cols_timed) == m, cols_timed
    if a.pmc_child:   # (the run live_traffic() counts: nothing but the iterations themselves)
        run.close()
        print(json.dumps({"pmc_child": True, "steps": a.steps, "ms_per_step": dt / a.steps * 1e3}))
        return
    stats, st0 = r["st1"], r["st0"]
    # ---- the timed path checks itself (every NEW_X row of this leg, warm-up and timed region) ----
    parity = parity_in_run(run.rows, n, m, a.real32, 1 if a.rosenbrock else 0)
    # (fatal at N = 1, where the run is the reference's run row for row; with several ranks the sums are added in
    #  another order and a late line search may legitimately take another trial -- reported, not fatal)
    # ... and only for the shapes whose rows the GPU tests pin as well (the long m = 32 / m = 48 / n = 1e6 runs go
    # 45-70 iterations, where another order of a reduction may take the walk one breakpoint further: reported)
    kind_ = 1 if a.rosenbrock else 0
    pinned_shape = (n, m, kind_) in ((100_000_000, 10, 0), (10_000_000, 10, 1), (100_000_000, 20, 0))
    parity_fail = parity["ok"] is False and world == 1 and pinned_shape
    # time to solution as the caller sees it: host clock from just before START to the NEW_X return of
    # iteration 30 (the two barriers of the timing protocol are inside; each costs a stream sync)
    tts30 = None
    for (it, *_), ts in zip(run.rows, run.stamps):
        if it == 30:
            tts30 = ts - run.t_start
    first_per_rank = [r["first_iter_s"]]
    if world > 1:
        ft = torch.tensor([r["first_iter_s"]], dtype=torch.float64, device=run.dev)
        fl = [torch.zeros_like(ft) for _ in range(world)]
        dist.all_gather(fl, ft)
        first_per_rank = [float(v[0]) for v in fl]
    f_final = float(sol.f[0])
    col = int(sol.isave[27])
    nfree = int(sol.isave[37])

    # ---- rooflines, live, hipEvents on the solver's stream ----
    head = int(sol.isave[26])
    mc = 5 if col <= 5 else 10 if col <= 10 else 20 if col <= 20 else 32
    # load policy the library picked (solver.hip init): nontemporal when W >> Infinity Cache
    ld_rows = (n_loc + 31) // 32 * 32
    nt = 2 * ld_rows * m * rbytes > (192 << 20)
    if "nt" in opts:
        nt = opts["nt"] != 0
    nts = "true" if nt else "false"
    tname = "float" if a.real32 else "double"
    lean = opts.get("lean", 1) != 0
    ub = sol.uniform_bounds()
    cst = sol.compact_stats()          # (packs, unpacks, packed now, eligible)
    wfrac = (nfree / float(n)) if (cst[3] and cst[2]) else 1.0   # (packed: the passes read the free rows' entries)
    upd_bpr, sub_bpr, n_st = pass_bytes(col, rbytes, run.pp, lean, ub, wfrac)
    upd_masked, sub_masked, _ = pass_bytes(col, rbytes, run.pp, lean, ub, 1.0)
    ubs = "_ub%d" % ub if ub else ""

    def static_traffic(fname, key, rows):
        tf = os.path.join(ROOT, "profiles", fname)
        if os.path.exists(tf) and not a.real32 and col == 10:   # measured for fp64, col = 10
            try:
                d = json.load(open(tf))
                d = d[key] if key else d
                return d.get("hbm_bytes_per_row") * rows
            except Exception:   # noqa: BLE001
                return None
        return None

    def in_run(name):
        ms, cnt = clocks[name]
        return (ms / cnt, cnt) if cnt else (None, 0)

    # Average launch duration over the timed region (in-run hipEvents); the same kernel launched
    # back to back after the run is kept beside it as a cross-check.  Passes over W of one
    # iteration: update_scan_kernel (read-only: the WS/WY matvecs W'd of cauchy and S'y, S's of
    # matupd, + formk's new row), subsm_update_kernel (W wv; the one pass that stores) and -- only
    # where the closed form for W'Z r does not apply (m > 20, long walks) -- cmprlb_wtv_kernel.
    def pass_record(key, which, label, bpr, stores, traffic_key):
        alg_bytes = bpr * n_loc
        ms_iso = sol.kernel_time(which, run.x, run.g, col, head, a.roofline_reps)
        ms_run, cnt = in_run(key)
        ms = ms_run if ms_run is not None else ms_iso
        ach = alg_bytes / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": label, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS,
                # (the counts on file are of the kernels that stream every row: not those of the compact layout)
                "traffic": None if (cst[3] and cst[2]) else static_traffic("w_pass_traffic.json", traffic_key, n_loc),
                "traffic_source": "profiles/w_pass_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                  "over this bench at n=1e8, in-iteration launches of this kernel: bytes per "
                                  "row x rows; not re-measured in this run)",
                "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_bytes_per_row": bpr,
                "avg_launch_ms": ms, "launches_timed": cnt,
                "timing": ("hipEvents around each launch inside the timed region" if cnt else
                           "not launched inside the timed region: back-to-back launches after it"),
                "avg_launch_ms_back_to_back": ms_iso, "rows_per_launch": n_loc, "col": col,
                "stores": stores}
    wide = m > 32   # beyond the width of the fused kernels: the iteration out of tile primitives
    if wide:
        # no fused pass to time: one record for the whole iteration, priced at the bytes the fused route would move
        it_bytes = (upd_bpr + sub_bpr + 2 * rbytes) * n_loc
        ach = it_bytes / (dt / a.steps) / 1e9
        wide_rec = {"bound": "hbm", "kernel": "m > 32: unfused tile passes (solver_wide.inl) -- the whole iteration, "
                    "priced at the bytes of the fused two-pass route", "achieved": ach, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                    "algorithmic_bytes_per_launch": it_bytes, "avg_launch_ms": dt / a.steps * 1e3,
                    "launches_timed": a.steps, "rows_per_launch": n_loc, "col": col}

    def pass_record_or_none(*args):
        return None if wide else pass_record(*args)
    rec_us = pass_record_or_none("update_scan", 4, "update_scan_kernel<%s, %d, %s> (run as the evaluation of the "
                         "first trial point: W'd of cauchy, S'y / S's of matupd, formk's new row%s)"
                         % (tname, mc, nts, "; col - 1 > 20: two launches of the <20> kernel over half of the columns "
                            "each + a merge, timed together, priced at the bytes of ONE pass" if mc > 20 else
                            ("; compact_w: lane pairs share the column sums, W entries of the free rows only"
                             if (cst[3] and cst[2]) else "")),
                         upd_bpr, "none", "update_scan" + ubs)
    entry = ("ping-pong entry" if run.pp else "classic entry") + (
        ", W in the tile-local free-row layout: option compact_w" if (cst[3] and cst[2]) else "")
    if run.pp:
        st_txt = "trial x + Ws/Wy column (3 of %d streams; t = x, r = g are a change of roles)" % (2 * col + 7)
    elif lean:
        st_txt = "t, r, trial x + Ws/Wy column (5 of %d streams)" % (2 * col + 9)
    else:
        st_txt = "z, d, t, r, trial x + Ws/Wy column (7 of %d streams)" % (2 * col + 11)
    rec_su = pass_record_or_none("subsm_update", 3, "subsm_update_kernel<%s, %d, %s> (%s, pending pair committed)"
                         % (tname, mc, nts, entry), sub_bpr, st_txt,
                         ("subsm_update_pp" if run.pp else "subsm_update") + ubs)
    rec_cw = pass_record_or_none("cmprlb_wtv", 2, "cmprlb_wtv_kernel<%s, %d, true, %s>" % (tname, mc, nts),
                         (2 * col + 2) * rbytes + 1, "none", "cmprlb_wtv")
    closed_steps, three_steps, handed_windows = sol.path_counts()
    if wide:
        roofline, others, roofline_wtv = wide_rec, [], None
        rec_us = {"avg_launch_ms": None, "frac": None}
    else:
      if not rec_cw["launches_timed"]:
        rec_cw["note"] = "not part of this run's iteration (W'Z r in closed form); timed back to back"
      # the headline roofline is the DOMINANT kernel of the iteration: the one with the largest total
      # time inside the timed region -- the storing pass
      recs = [rec_su, rec_us] + ([rec_cw] if rec_cw["launches_timed"] else [])
      recs.sort(key=lambda d: -(d["avg_launch_ms"] * max(d["launches_timed"], 1)))
      roofline = recs[0]
      roofline["dominant"] = "largest share of the iteration: %.2f of %.2f ms per step" % (
        roofline["avg_launch_ms"] * roofline["launches_timed"] / a.steps, dt / a.steps * 1e3)
      others = recs[1:] + ([] if rec_cw["launches_timed"] else [rec_cw])
      ms_kernel = sol.wtv_time(run.g, col, head, a.roofline_reps)
      alg_bytes = (2 * col + 1) * n_loc * rbytes
      achieved = alg_bytes / (ms_kernel * 1e-3) / 1e9
      roofline_wtv = {"bound": "hbm", "kernel": "wtv_kernel<%s, %d, %s> (the bare WS/WY matvec W'v)" % (tname, mc, nts),
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": static_traffic("wtv_traffic.json", None, n_loc),
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
        "warmup_run": r["warm_done"],
        "col_min_timed": min(cols_timed),
        "col_max_timed": max(cols_timed),
        "ms_per_step": dt / a.steps * 1e3,
        "ms_per_step_median": r["step_ms_median"],
        "ms_per_step_max": r["step_ms_max"],
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32" if a.real32 else "f64",
        "data": "synthetic",
        "config": {"workload": ("extended Rosenbrock with the drivers' box (--rosenbrock), n=%d, m=%d, %s, on-device "
                                "objective" if a.rosenbrock else
                                "separable bounded quadratic (SURVEY.md 8d), n=%d, m=%d, %s, "
                                "l=-1,u=1,x0=0, on-device objective")
                               % (n, m, "fp32 storage/fp64 accumulate" if a.real32 else "fp64"),
                   "n": n, "m": m, "rows_per_gpu": n_loc, "parallelism": "rows/%d" % world,
                   "collective": run.collective,
                   "entry": ("lbfgsb_hip_setulb_dev_pp (ping-pong iterate buffers)" if run.pp
                             else "lbfgsb_hip_setulb_dev"),
                   "options": opts,
                   "lnsrch_setup": ("waited for at every FG_LNSRCH return (--no-defer)" if a.no_defer else
                                    "LBFGSB_F_DEFER_LNSRCH: the objective is the library's own kernel on the "
                                    "solver's stream, so the storing pass's sums ride with the next call's fetch "
                                    "(one host sync per iteration less; NEW_X returns bit-identical)"),
                   # (short values: the driver's record keeps this dict, long strings are cut)
                   "parity_in_run": parity,
                   "rccl_nranks": (sol.comm_info()[0] if sol.comm_info()[2] == 1 else None),
                   "first_iteration_s": r["first_iter_s"],
                   "first_iteration_s_per_rank": first_per_rank,
                   "refresh_count": stats["refreshes"],
                   "time_to_30_iterations_s": tts30,
                   "iterations_run": run.rows[-1][0] if run.rows else 0,
                   "defer_lnsrch": not a.no_defer,
                   # option compact_w: the passes over W on the tile-local free-row layout -- [on, tile re-sorts,
                   # un-sorts, packed at the end, fraction of the rows whose W entries a pass needs (nfree / n)]
                   "compact_w": [cst[3], cst[0], cst[1], cst[2], round(wfrac, 4)],
                   "bytes_per_row_masked_design": [upd_masked, sub_masked],
                   "uniform_bounds_mask": ub,
                   "uniform_bounds": ("l, u few-valued: dictionary-coded in the nbd byte (bit 3)" if ub & 8 else
                                      "l, u, nbd hold one value each: read as constants (bits 0-2); leg ub_off streams them"
                                      if ub else "not detected / off: l, u, nbd are streamed")},
        "iters_per_sec_setulb_only": a.steps / dt_setulb,
        "first_iteration_s": r["first_iter_s"],
        "first_iteration_nseg": r["nseg_first"],
        "f_final": f_final,
        "col": col,
        "nfree": nfree,
        "host_syncs_per_iter": (stats["syncs"] - st0["syncs"]) / a.steps,
        "collectives_per_iter": (stats["collectives"] - st0["collectives"]) / a.steps,
        # median / minimum of 1000 host syncs by themselves on this communicator, measured before START
        "collective_us": coll_us[0] if coll_us else None,
        "collective_us_min": coll_us[1] if coll_us else None,
        # what the communicator itself says (ncclCommCount / ncclCommUserRank): N ranks took part
        "rccl_nranks": (sol.comm_info()[0] if sol.comm_info()[2] == 1 else None),
        "comm_kind": {0: "none", 1: "rccl", 2: "host callbacks"}[sol.comm_info()[2]],
        # slowest / fastest rank over the timed region (value is K / the slowest)
        "ms_per_step_rank_max": dt / a.steps * 1e3,
        "ms_per_step_rank_min": r["dt_min"] / a.steps * 1e3,
        "kernel_launches_per_iter": (stats["launches"] - st0["launches"]) / a.steps,
        "host_blocked_ms_per_iter": (stats["wait_seconds"] - st0["wait_seconds"]) / a.steps * 1e3,
        "cauchy_fullsorts": stats["cauchy_fullsorts"],
        "freev_passes_skipped_per_iter": (stats["freev_skipped"] - st0["freev_skipped"]) / a.steps,
        "tie_splits": sol.tie_splits(),
        "lnsrch_setups_deferred_reissued": list(sol.defer_stats()),
        "host_algebra_us_between_passes": stats["host_gap_us"],
        "host_segments_us": stats["host_segments_us"],
        "subspace_steps_closed_form": closed_steps,
        "subspace_steps_three_pass": three_steps,
        "cauchy_walks_served_by_update_pass": handed_windows,
        # the metric's own kernel, in-run (hipEvents around its launches inside the timed region): the
        # read-only pass that carries the WS/WY matvecs of the iteration (W'd of cauchy, S'y / S's of matupd)
        "update_scan_ms_in_run": rec_us["avg_launch_ms"],
        "update_scan_frac_of_hbm_peak": rec_us["frac"],
        # the WHOLE iteration against the roofline: algorithmic bytes of one iteration (the two passes over
        # W + the caller's objective kernel: x read, g written) / ms_per_step
        "roofline_iteration": {
            "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "algorithmic_bytes_per_iteration": (upd_bpr + sub_bpr + 2 * rbytes) * n_loc,
            "achieved": (upd_bpr + sub_bpr + 2 * rbytes) * n_loc / (dt / a.steps) / 1e9,
            "frac": (upd_bpr + sub_bpr + 2 * rbytes) * n_loc / (dt / a.steps) / 1e9 / HBM_PEAK_GBS,
            "note": "per rank; window / freev / gather kernels of the iterations that need them are not in the "
                    "numerator, so this is a lower bound of the bytes moved"},
        "roofline": roofline,
        "roofline_wtv": roofline_wtv,
        "roofline_other_w_passes": others,
    }
    run.close()
    # the opt-in closed-form GCP (LBFGSB_F_PARALLEL_GCP): only the first iteration differs
    # (col = 0, nseg ~ 0.977 n); timed on a fresh context, outside the timed region above
    try:
        if world > 1:      # single-GPU leg only: no second communicator inside the scaling runs
            raise RuntimeError("skipped for n_gpus > 1")
        run2 = Run(torch, dist, lbfgsb_amd, a, n=n, m=m, real32=a.real32, kind=0, world=1, rank=0,
                   local_rank=local_rank, rccl_self=False, opts=opts, parallel_gcp=True)
        run2.barrier()
        tp0 = time.perf_counter()
        run2.advance(1)
        run2.barrier()
        out["first_iteration_parallel_gcp_s"] = time.perf_counter() - tp0
        out["first_iteration_parallel_gcp_nseg"] = int(run2.sol.isave[32])
        run2.close()
    except Exception as e:   # noqa: BLE001
        out["first_iteration_parallel_gcp_s"] = None
        out["first_iteration_parallel_gcp_error"] = repr(e)
    # ---- the other BASELINE.json configs, short legs (N = 1 only; each a fresh context) ----
    if world == 1 and not a.no_other_configs and not a.real32 and n == 100_000_000 and m == 10:
        leg_tags = ["ordinary_caller", "torch_caller", "ub_off", "cfg1_n1e6", "cfg2_rosen_n1e7", "cfg4_r32_m20", "cfg3_rank_shape",
                    "cfg3_rank_shape_nodefer", "m32_n5e7", "m48_n2e7"]
        leg_defs = [
            ("ORDINARY CALLER, ARBITRARY BOX: the headline problem size with per-variable bounds (every l_i, u_i its own "
             "value: l, u, nbd streamed by the passes, no constants, no dictionary), the classic in-place entry "
             "lbfgsb_hip_setulb_dev (t = x, r = g as copies) and no LBFGSB_F_DEFER_LNSRCH -- nothing the headline leans on",
             dict(n=n, m=m, real32=False, kind=0, rccl_self=False, steps=20, warm_min=12, defer=False, classic=True,
                  arbitrary_box=True)),
            ("TORCH-STREAM CALLER, ARBITRARY BOX: the same per-variable bounds, the objective evaluated by torch ops on "
             "torch's current stream and ordered with events (stream_ordered: lbfgsb_hip_return_event / _wait_stream), "
             "f handed over as a device scalar (lbfgsb_hip_f_device), ping-pong entry + LBFGSB_F_DEFER_LNSRCH -- the "
             "deferred path for a caller that is NOT on the solver's stream",
             dict(n=n, m=m, real32=False, kind=0, rccl_self=False, steps=20, warm_min=12, defer=True, classic=False,
                  arbitrary_box=True, torch_objective=True)),
            ("headline workload with the uniform-bounds detection OFF (l, u, nbd streamed per row)",
             dict(n=n, m=m, real32=False, kind=0, rccl_self=False, steps=20, warm_min=12,
                  opts=dict(opts, uniform_bounds=0))),
            ("configs[1]: separable bounded quadratic n=1e6, m=10, fp64", dict(n=1_000_000, m=10, real32=False,
             kind=0, rccl_self=False, steps=40, warm_min=12)),
            ("configs[2]: extended Rosenbrock with box bounds n=1e7, m=10, fp64", dict(n=10_000_000, m=10,
             real32=False, kind=1, rccl_self=False, steps=16, warm_min=12)),
            ("configs[4]: n=1e8, m=20, REAL32 (fp32 storage/kernels, fp64 accumulators)", dict(n=100_000_000,
             m=20, real32=True, kind=0, rccl_self=False, steps=16, warm_min=21)),
            ("configs[3] per-rank shape: 1.25e7 rows (n=1e8 over 8 GPUs), 1-rank RCCL communicator behind "
             "every sync", dict(n=12_500_000, m=10, real32=False, kind=0, rccl_self=True, steps=60, warm_min=12)),
            ("the same per-rank shape WITHOUT LBFGSB_F_DEFER_LNSRCH (every FG_LNSRCH return waits for the storing "
             "pass's sums: what an ordinary reverse-communication caller gets)",
             dict(n=12_500_000, m=10, real32=False, kind=0, rccl_self=True, steps=60, warm_min=12, defer=False)),
            ("m = 32 (two passes over W per iteration; the update pass split over the columns: col > 21), n = 5e7, fp64", dict(n=50_000_000, m=32,
             real32=False, kind=0, rccl_self=False, steps=10, warm_min=33)),
            ("m = 48 (> 32 pairs: the update pass split over the columns in front of the unfused subspace steps, W'Z r in "
             "closed form, one axpy pass -- DESIGN.md 4f), n = 2e7, fp64", dict(n=20_000_000, m=48, real32=False, kind=0,
             rccl_self=False, steps=10, warm_min=49)),
        ]
        out["other_configs"] = []
        for name, kw in leg_defs:
            try:
                kw.setdefault("opts", opts)
                out["other_configs"].append(other_config(torch, dist, lbfgsb_amd, a, name, local_rank=local_rank,
                                                         **kw))
            except Exception as e:   # noqa: BLE001  (a leg must never take the headline line down; Ctrl-C still ends the run)
                out["other_configs"].append({"config": name, "error": repr(e)})
        # the host-pointer form (reference argument list, host arrays): PCIe-inclusive, never the headline
        try:
            out["host_form"] = host_form_leg(lbfgsb_amd)
            parity_fail = parity_fail or not out["host_form"]["anchors_ok"]
        except Exception as e:   # noqa: BLE001
            out["host_form"] = {"error": repr(e)}
        # <= 1 KB summary of every leg INSIDE config (the driver's record keeps config; other_configs is long)
        legs = {}
        for tag, oc in zip(leg_tags, out["other_configs"]):
            if "error" in oc:
                legs[tag] = "error " + oc["error"][:60]
                continue
            ps = oc["passes"]
            fr = lambda k: ("%.2f" % ps[k]["frac"]) if k in ps else "-"     # noqa: E731
            legs[tag] = "%.1f it/s %.3f ms upd %s sub %s syncs %.2f" % (
                oc["value"], oc["ms_per_step"], fr("update_scan"), fr("subsm_update"), oc["host_syncs_per_iter"])
            if oc.get("collective_us") is not None:
                legs[tag] += " coll %.0f us" % oc["collective_us"]
            pr = oc.get("parity_in_run") or {}
            if pr.get("rows_checked"):
                legs[tag] += (" parity %d rows ok" % pr["rows_checked"] if pr["ok"] else
                              " parity %d/%d rows, first split" % (pr["rows_ok_before_first_split"], pr["rows_checked"]))
                # (fatal for the shapes whose rows the GPU tests pin as well: the headline's problem, Rosenbrock,
                #  config 5; the long m = 32 / m = 48 legs and n = 1e6 run 45-70 iterations, where a reduction
                #  order may legitimately take another line-search trial: reported)
                if tag in ("ub_off", "cfg2_rosen_n1e7", "cfg4_r32_m20"):
                    parity_fail = parity_fail or not pr["ok"]
        oc0 = out["other_configs"][0]
        out["iters_per_sec_ordinary_caller_arbitrary_box"] = oc0.get("value")
        out["config"]["ordinary_caller_arbitrary_box_its"] = oc0.get("value")
        hf = out["host_form"]
        legs["host_form_n1e7"] = ("error " + hf["error"][:60]) if "error" in hf else (
            "%.1f it/s %.3f ms in setulb, PCIe %.1f GB/s, anchors %s" % (
                hf["value"], hf["ms_per_step"], hf["pcie_GBs_inside_setulb"], "ok" if hf["anchors_ok"] else "MISMATCH"))
        out["config"]["legs"] = legs
    # ---- HBM traffic of the passes over W, counted in this run (contexts above are closed: the child
    # has the card's memory to itself) ----
    if rank == 0 and world == 1 and not a.no_live_traffic:
        static_note = ("profiles/w_pass_traffic.json (rocprofv3 --pmc passes over this bench on file; the live "
                       "count of this run was not taken: %s)")
        try:
            if under_profiler():
                raise RuntimeError("this process itself runs under a profiler")
            lt = live_traffic(a, n_loc)
            for rec in [roofline] + others:
                fam = rec["kernel"].split("<", 1)[0]
                if fam in lt:
                    rec["traffic_on_file"] = rec["traffic"]
                    rec["traffic"] = lt[fam]["hbm_bytes_per_launch"]
                    rec["traffic_over_algorithmic"] = rec["traffic"] / rec["algorithmic_bytes_per_launch"]
                    rec["traffic_source"] = (
                        "counted in this run: two child runs of this bench (same workload, same entry) under "
                        "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE with --kernel-trace, launches at col = m; "
                        "gfx950 corrections of MI355X_MICROARCH.md (KiB units, FETCH_SIZE x2)")
                    rec["traffic_counters"] = lt[fam]
            out["live_traffic_seconds"] = lt["seconds"]
        except Exception as e:   # noqa: BLE001  (the counters must never take the bench line down; Ctrl-C still ends the run)
            out["live_traffic_error"] = repr(e)[:400]
            for rec in [roofline] + others:
                rec["traffic_source"] = static_note % repr(e)[:120]
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(m, n, min(a.cpu_n, n))
        except Exception as e:  # the baseline must never take the bench line down
            out["cpu_baseline"] = {"value": None, "unit": "iters/sec", "cores": 1, "kind": "reference",
                                   "sample": "failed: %r" % (e,)}
    if rank == 0:
        emit(out)
    if world > 1:
        dist.destroy_process_group()
    if parity_fail:      # (the line above is on record; the run itself must not pass)
        sys.stderr.write("bench.py: parity_in_run FAILED: %s\n" % json.dumps(parity))
        raise SystemExit(3)


if __name__ == "__main__":
    main()

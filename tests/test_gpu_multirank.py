"""The N>1 path on the one GPU a test box has: (a) two ranks sharing cuda:0 with a gloo
host group completing the reductions and the breakpoint all-gather, (b) an RCCL
communicator of ONE rank, which drives ncclAllReduce/ncclAllGather on the solver's stream.
Both must reproduce the single-rank oracle trajectory: integers exactly, f to 1e-9."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(world, mode, n, m, iters, mixed, out):
    port = free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_mr_worker.py"), str(r), str(world),
                               str(port), mode, str(n), str(m), str(iters),
                               mixed if isinstance(mixed, str) else ("1" if mixed else "0"), out])
             for r in range(world)]
    rcs = [p.wait(timeout=300) for p in procs]
    assert rcs == [0] * world, rcs
    return json.load(open(out))


def oracle_rows(po, n, m, iters, mixed):
    if mixed == "rosen":
        p = po.problem_rosenbrock(n, m, factr=0.0, pgtol=0.0)
    else:
        p = po.problem_quadratic(n, m, mixed_nbd=(mixed is True))
    rows = []
    s = po.run(po.Engine("oracle"), p, max_iter=iters,
               snapshot=lambda k, s: rows.append([int(s.isave[29]), int(s.isave[33]), int(s.isave[32]),
                                                  int(s.isave[37]), float(s.f[0]), float(s.dsave[12])])
               if s.task_s.startswith("NEW_X") else None)
    return rows, s.x.copy()


@pytest.mark.parametrize("world,mode,n,m,iters,mixed", [
    (2, "gloo", 20011, 7, 8, True),       # ragged split, all four bound types
    (3, "gloo", 300000, 10, 3, False),    # iteration 1 walks ~293k breakpoints: full sort + merged chunks
    (4, "gloo", 10007, 5, 6, True),       # four ranks on one GPU, m = 5
    (2, "gloo", 30011, 20, 26, True),     # m = 20 until the memory is full: the pair-shared update pass,
                                          # its leftover rows, the closed form at col = 20, on 2 ranks
    (2, "gloo", 20011, 27, 33, True),     # m = 27: the update pass split over the columns (col - 1 > 20), its two
                                          # result sets merged before the ranks' sums are gathered
    (1, "rccl1", 50021, 5, 6, True),
    (3, "gloo", 6007, 40, 46, True),      # m > 32: the unfused tile path (solver_wide.inl) over 3 ranks -- WN1 kept
                                          # incrementally, the changed rows' patch summed per rank and reduced
    (5, "fakerccl", 500009, 10, 14, False),  # FIVE rank PROCESSES (+ this one: the six a GPU box admits on its card) through
                                          # the communicator code path with the ASYNCHRONOUS stand-in for librccl: the
                                          # headline problem's shape at 1e5 rows per rank -- a first walk of ~488 k
                                          # breakpoints all-gathered in chunks and merged on the device of every rank,
                                          # the filling memory, one all-gather per host sync (the 8-rank full-size run of
                                          # test_gpu_config4.py has to use rank threads and the synchronous stand-in)
])
def test_sharded_trajectory_matches_oracle(oracle_built, tmp_path, monkeypatch, world, mode, n, m, iters, mixed):
    po = oracle_built
    if mode == "fakerccl":
        monkeypatch.setenv("LBFGSB_RCCL_LIBRARY", _fake_rccl())
    res = launch(world, mode, n, m, iters, mixed, str(tmp_path / "out.json"))
    rows, x = oracle_rows(po, n, m, iters, mixed)
    assert len(res["rows"]) == len(rows) == iters
    for a, b in zip(res["rows"], rows):
        assert a[:4] == b[:4], (a, b)                      # iter nfgv nseg nfree
        assert a[4] == pytest.approx(b[4], rel=1e-9)       # f
        assert a[5] == pytest.approx(b[5], rel=1e-7)       # |proj g|
    xa = np.array(res["x"])
    assert np.max(np.abs(xa - x)) <= 1e-8 * max(1.0, np.max(np.abs(x)))
    if n >= 300000:
        assert res["stats"]["cauchy_fullsorts"] >= 1


def test_sharded_rosenbrock_halo_objective(oracle_built, tmp_path):
    """Extended Rosenbrock (test/driver1.f90:272-291 formulas, driver bounds) with the rows cut
    over 3 ranks: the built-in objective needs x(i-1), x(i+1) across the cuts (1-element halo,
    all-gathered).  Same trajectory as the single-rank oracle."""
    po = oracle_built
    n, m, iters = 30011, 5, 10
    res = launch(3, "gloo", n, m, iters, "rosen", str(tmp_path / "out.json"))
    rows, x = oracle_rows(po, n, m, iters, "rosen")
    assert len(res["rows"]) == len(rows) == iters
    for a, b in zip(res["rows"], rows):
        assert a[:4] == b[:4], (a, b)
        assert a[4] == pytest.approx(b[4], rel=1e-9)
    xa = np.array(res["x"])
    assert np.max(np.abs(xa - x)) <= 1e-8 * max(1.0, np.max(np.abs(x)))


def test_sharded_parallel_gcp(oracle_built, tmp_path):
    """The opt-in closed-form GCP with the rows cut over 2 ranks (one summed count, no
    breakpoint exchange): nseg within 2 of the oracle, f to 1e-9, no sort anywhere."""
    po = oracle_built
    n, m, iters = 200003, 10, 3
    res = launch(2, "gloo", n, m, iters, "pgcp", str(tmp_path / "out.json"))
    rows, x = oracle_rows(po, n, m, iters, False)
    assert len(res["rows"]) == len(rows) == iters
    for a, b in zip(res["rows"], rows):
        assert a[:2] == b[:2], (a, b)
        assert abs(a[2] - b[2]) <= 2 and abs(a[3] - b[3]) <= 2, (a, b)
        assert a[4] == pytest.approx(b[4], rel=1e-9)
    assert res["stats"]["cauchy_fullsorts"] == 0


@pytest.mark.parametrize("world,mode", [(3, "gloo"), (2, "fakerccl")])
def test_sharded_parallel_gcp_with_pairs_stored(oracle_built, tmp_path, monkeypatch, world, mode):
    """The parallel GCP search with pairs stored (col > 0) over several ranks: every rank sorts and
    gathers the records of its own breakpoints, the records are all-gathered (host callback / the
    library's ncclAllGather path), merged by (t, global index), and each rank runs the scans on
    all of them.  Same two-scale problem as the single-rank test: iterations 2 and 6 cross
    ~ 91 000 breakpoints with col = 1 and col = 5; nseg / nfree within 2 of the oracle's
    sequential walk, f to 1e-9, two full sorts per rank."""
    po = oracle_built
    import importlib.util
    spec = importlib.util.spec_from_file_location("_mr_worker", os.path.join(HERE, "_mr_worker.py"))
    wk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(wk)
    n, m, iters = 200_000, 5, 8
    if mode == "fakerccl":
        monkeypatch.setenv("LBFGSB_RCCL_LIBRARY", _fake_rccl())
    res = launch(world, mode, n, m, iters, "pgcp2", str(tmp_path / "out.json"))
    p, _ = wk.two_scale_problem(po, n, m)
    rows = []
    po.run(po.Engine("oracle"), p, max_iter=iters,
           snapshot=lambda k, s: rows.append([int(s.isave[29]), int(s.isave[33]), int(s.isave[32]),
                                              int(s.isave[37]), float(s.f[0])])
           if s.task_s.startswith("NEW_X") else None)
    assert sum(1 for r in rows if r[2] > 50_000) >= 2
    assert len(res["rows"]) == len(rows) == iters
    for a, b in zip(res["rows"], rows):
        assert a[:2] == b[:2], (a, b)
        assert abs(a[2] - b[2]) <= 2 and abs(a[3] - b[3]) <= 2, (a, b)
        assert a[4] == pytest.approx(b[4], rel=1e-9)
    assert res["stats"]["cauchy_fullsorts"] >= 2


@pytest.mark.parametrize("world,mode", [(3, "gloo"), (2, "fakerccl")])
def test_sharded_exact_tie_order(oracle_built, tmp_path, monkeypatch, world, mode):
    """The reference's tie order over several ranks: every rank holds all breakpoint times, pops the
    same replicated heap (hpsolb, src/lbfgsb.f90:2079-2157) and gathers the records of the rows it
    owns.  On a problem made of 8 exact copies of 257 variables -- groups of equal breakpoints
    whose members sit on different ranks, alive over 40 iterations -- the iterate x itself must
    equal the single-rank oracle's (not only up to a permutation inside the groups, which is what
    the default variable order gives)."""
    po = oracle_built
    import importlib.util
    spec = importlib.util.spec_from_file_location("_mr_worker", os.path.join(HERE, "_mr_worker.py"))
    wk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(wk)
    n, m, iters = 257 * 8, 5, 40
    if mode == "fakerccl":
        monkeypatch.setenv("LBFGSB_RCCL_LIBRARY", _fake_rccl())
    p, _ = wk.symmetric_problem(po, n, m)
    rows = []
    so = po.run(po.Engine("oracle"), p, max_iter=iters,
                snapshot=lambda k, s: rows.append([int(s.isave[29]), int(s.isave[33]), int(s.isave[32]),
                                                   int(s.isave[37]), float(s.f[0])])
                if s.task_s.startswith("NEW_X") else None)
    # (on this trajectory few walks happen to END inside a group, so the replay would hardly be
    #  triggered by itself: variant "symx" sets the option exact_always, which sends every walk
    #  through the multi-rank heap order)
    res = launch(world, mode, n, m, iters, "symx", str(tmp_path / "out.json"))
    assert res["stats"]["syncs"] > 0
    assert len(res["rows"]) == len(rows) == iters
    for a, b in zip(res["rows"], rows):
        assert a[:4] == b[:4], (a, b)
        assert a[4] == pytest.approx(b[4], rel=1e-10)
    assert np.max(np.abs(np.array(res["x"]) - so.x)) <= 1e-9
    # variable order (LBFGSB_F_INDEX_TIES) on the same ranks: same scalars, x equal up to a
    # permutation inside the groups
    res_d = launch(world, mode, n, m, iters, "sym", str(tmp_path / "out2.json"))
    for a, b in zip(res_d["rows"], rows):
        assert a[:4] == b[:4], (a, b)
    xa = np.sort(np.array(res_d["x"]).reshape(8, 257), axis=0)
    xb = np.sort(so.x.reshape(8, 257), axis=0)
    assert np.max(np.abs(xa - xb)) <= 1e-9
    print("tie splits: %d (variable order), %d (heap order)" % (res_d["tie_splits"], res["tie_splits"]))


@pytest.mark.parametrize("world,mode", [(3, "gloo"), (2, "fakerccl")])
def test_sharded_checkpoint_and_resume(oracle_built, tmp_path, monkeypatch, world, mode):
    """export_state / import_state on a sharded run: at iteration 7 of 14 every rank exports its rows
    (reference wa / iwa layout with n = n_local; Indx2(1) = local free count), the contexts are
    destroyed and rebuilt, the state is imported and the run goes on -- the trajectory must be the
    uninterrupted one (= the single-rank oracle's): integer columns exactly, f to 1e-9."""
    po = oracle_built
    n, m, iters = 40_009, 6, 14
    if mode == "fakerccl":
        monkeypatch.setenv("LBFGSB_RCCL_LIBRARY", _fake_rccl())
    res = launch(world, mode, n, m, iters, "ckpt", str(tmp_path / "out.json"))
    rows, x = oracle_rows(po, n, m, iters, False)
    assert len(res["rows"]) == len(rows) == iters
    for a, b in zip(res["rows"], rows):
        assert a[:4] == b[:4], (a, b)
        assert a[4] == pytest.approx(b[4], rel=1e-9)
    assert np.max(np.abs(np.array(res["x"]) - x)) <= 1e-8 * max(1.0, np.max(np.abs(x)))


@pytest.mark.parametrize("world,first,count", [(2, 100, 25), (3, 200, 25)])
def test_sharded_random_problems_match_oracle(oracle_built, tmp_path, world, first, count):
    """Random separable problems (n < 2000, m < 13, all bound types) with the rows cut over 2 and
    3 ranks, the objective evaluated per shard and summed: call by call beside the single-rank oracle's
    trajectory -- the sharded reductions, the merged breakpoint walk, the fix lists and the pending pair
    all have to agree on every rank for that.  Where a sharded run leaves the trajectory (rounding drift:
    factr = 0), the ranks' exported rows before and after that call are put together and ONE oracle call
    from the run's own previous state must reproduce the call (task, every counter, iwhere exactly, floats
    to 1e-10): the bar of tests/test_gpu_fuzz.py, no allowance."""
    iters = 40
    res = launch(world, "fuzz", first, count, iters, "-", str(tmp_path / "out.json"))
    splits = 0
    for seed in range(first, first + count):
        r = res[str(seed)]
        if r["split"] is None:
            assert r["calls"] == r["oracle_calls"], (seed, r["calls"], r["oracle_calls"])
        else:
            splits += 1
            assert r["verdict"] == "reproduced", (seed, r["verdict"])
    assert splits <= 0.2 * count, splits   # (a sanity bound on how often drift shows, not an allowance)


def _fake_rccl():
    """Build tests/fake_rccl.cpp (shared-memory stand-in for the RCCL entry points) on demand."""
    so = os.path.join(HERE, "_build", "libfake_rccl.so")
    src = os.path.join(HERE, "fake_rccl.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        # (compiled as HIP: the stand-in launches one tiny kernel -- the stream-side wait of a collective)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-x", "hip", "--offload-arch=gfx950", "-O2", "-std=c++17",
                               "-fPIC", "-shared", src, "-o", so, "-lrt", "-lpthread"])
    return so


@pytest.mark.parametrize("world,n,m,iters,mixed", [
    (2, 20011, 7, 8, True),
    (3, 300000, 10, 3, False),     # long first walk: all-gathered record chunks, merged on every rank
    (4, 10007, 5, 6, "rosen"),     # halo exchange of the sharded objective through ncclAllGather
    (2, 10007, 27, 33, True),      # the split update pass: several launches, one merged result set gathered
    (2, 6007, 40, 46, True),       # m > 32: results wider than the fused layout, the changed-row patch reduced
])
def test_communicator_code_path_with_several_ranks(oracle_built, tmp_path, monkeypatch, world, n, m,
                                                   iters, mixed):
    """The solver's RCCL code path (lbfgsb_hip_comm_init_rccl; grouped sum/min/max ncclAllReduce of
    the partials, ncclAllGather of breakpoint records and halos, all on the solver's stream) with
    2-4 ranks.  Real RCCL refuses ranks that share a GPU, so LBFGSB_RCCL_LIBRARY points the
    library at a shared-memory stand-in for those seven entry points; everything above them is
    the production code.  Same trajectory as the single-rank oracle."""
    po = oracle_built
    monkeypatch.setenv("LBFGSB_RCCL_LIBRARY", _fake_rccl())
    res = launch(world, "fakerccl", n, m, iters, mixed, str(tmp_path / "out.json"))
    rows, x = oracle_rows(po, n, m, iters, mixed)
    assert len(res["rows"]) == len(rows) == iters
    for a, b in zip(res["rows"], rows):
        assert a[:4] == b[:4], (a, b)
        assert a[4] == pytest.approx(b[4], rel=1e-9)
    xa = np.array(res["x"])
    assert np.max(np.abs(xa - x)) <= 1e-8 * max(1.0, np.max(np.abs(x)))


def test_bench_two_ranks_rehearsal(tmp_path):
    """bench.py's multi-rank flow (row partition, communicator set-up with the id broadcast,
    barriers, MAX over ranks of the timings, rank-0 JSON line) on ONE GPU: LBFGSB_BENCH_SHARE_GPU puts both
    ranks on cuda:0 with a gloo group, LBFGSB_RCCL_LIBRARY swaps librccl for the stand-in.  Launched BARE --
    `python bench.py --gpus 2`, no launcher around it: bench.py starts torch.distributed.run itself as a child
    process (before it touches the GPU) and forwards rank 0's line and the exit code; the launched ranks take
    the very route the driver's own `python -m torch.distributed.run ... bench.py --gpus N` takes."""
    root = os.path.dirname(HERE)
    env = dict(os.environ, LBFGSB_BENCH_SHARE_GPU="1", LBFGSB_RCCL_LIBRARY=_fake_rccl(),
               MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rows", "2000000", "--steps", "5",
           "--warmup", "12", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert len(lines[0]) <= 4096                     # the compact line the driver parses (bench.LINE_CAP)
    out = json.loads(lines[0])
    cfg = out["config"]
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["value"] > 0
    assert cfg["rows_per_gpu"] == 1000000 and cfg["comm_kind"] == "rccl"
    assert cfg["collectives_per_iter"] == cfg["host_syncs_per_iter"] > 0     # one collective per host sync
    assert out["roofline"]["frac"] > 0 and out["scaling"] == "strong"
    # what a real scaling run needs to explain itself: the communicator's own rank count, the cost of
    # one host sync on it, the syncs per iteration, the first iteration and the timed region per rank
    assert cfg["rccl_nranks"] == 2 and len(cfg["first_iteration_s_per_rank"]) == 2
    assert cfg["collective_us"] > 0 and cfg["ms_per_step_rank_min"] <= cfg["ms_per_step_rank_max"]
    assert cfg["parity_in_run"]["rows_checked"] == 0     # (no reference rows for this small shape)
    detail = json.load(open(os.path.join(root, "bench_detail.json")))
    assert detail["first_iteration_nseg"] > 1900000     # ~0.977 n segments, walked across both ranks
    assert "RCCL" in detail["config"]["collective"]


@pytest.mark.parametrize("world,mode,n,m,iters,mixed", [
    (2, "fakerccl", 20011, 7, 12, True),      # the communicator code path: one all-gather carries both segments
    (3, "fakerccl", 10007, 5, 10, "rosen"),   # (+ the halo exchange of the sharded objective in between)
    (2, "gloo", 20011, 7, 12, True),          # host callbacks: the deferred segment is a second reduction
])
def test_deferred_line_search_sums_over_several_ranks(oracle_built, tmp_path, monkeypatch, world, mode, n, m,
                                                      iters, mixed):
    """LBFGSB_F_DEFER_LNSRCH with several ranks: the storing pass's four sums stay on every rank's device and are
    all-gathered / reduced with the NEXT fetch (solver.hip fetch, DEFER_OFF) -- every rank must take the same
    decisions from them.  Same NEW_X rows as the run without the flag, bit for bit (f and |proj g| included), and
    the oracle's trajectory."""
    po = oracle_built
    if mode == "fakerccl":
        monkeypatch.setenv("LBFGSB_RCCL_LIBRARY", _fake_rccl())
    base = launch(world, mode, n, m, iters, mixed, str(tmp_path / "base.json"))
    monkeypatch.setenv("LBFGSB_TEST_DEFER", "1")
    res = launch(world, mode, n, m, iters, mixed, str(tmp_path / "defer.json"))
    assert base["defer"] == [0, 0] and res["defer"][0] >= iters - 2, (base["defer"], res["defer"])
    assert res["rows"] == base["rows"]          # bit for bit: the same sums in the same order
    assert res["x"] == base["x"]
    rows, _ = oracle_rows(po, n, m, iters, mixed)
    assert len(res["rows"]) == len(rows) == iters
    for a, b in zip(res["rows"], rows):
        assert a[:4] == b[:4], (a, b)
        assert a[4] == pytest.approx(b[4], rel=1e-9)
    # one collective per host sync still holds, and there is one sync per iteration less
    assert res["stats"]["syncs"] < base["stats"]["syncs"]


@pytest.mark.parametrize("world,mode,defer", [(2, "gloo", False), (2, "fakerccl", True), (3, "fakerccl", False)])
def test_stop_cpu_sharded(oracle_built, tmp_path, monkeypatch, world, mode, defer):
    """task = 'STOP: CPU' (src/lbfgsb.f90:565-573, test/driver3.f90:151-182) on a sharded run, sent at the first trial
    point after iteration 9 (with LBFGSB_F_DEFER_LNSRCH: while that line search's set-up is still deferred): every rank
    gets its rows of the iterate and its gradient back bit for bit, f = dsave(2) = the iterate's value, and the rows
    up to there are the oracle's."""
    po = oracle_built
    n, m, iters = 20011, 7, 9
    if mode == "fakerccl":
        monkeypatch.setenv("LBFGSB_RCCL_LIBRARY", _fake_rccl())
    if defer:
        monkeypatch.setenv("LBFGSB_TEST_DEFER", "1")
    res = launch(world, mode, n, m, iters, "stopcpu", str(tmp_path / "out.json"))
    assert res["stopcpu"] is True
    assert res["task"].startswith("STOP: CPU")
    rows, x = oracle_rows(po, n, m, iters, False)
    assert len(res["rows"]) == len(rows) == iters
    for a, b in zip(res["rows"], rows):
        assert a[:4] == b[:4], (a, b)
    assert np.max(np.abs(np.array(res["x"]) - x)) <= 1e-9 * max(1.0, np.max(np.abs(x)))
    if defer:
        assert res["defer"][0] >= iters - 2


@pytest.mark.parametrize("world,mode,variant", [(2, "gloo", True), (3, "fakerccl", "ckpt")])
def test_sharded_runs_on_the_compact_layout(oracle_built, tmp_path, monkeypatch, world, mode, variant):
    """option compact_w on every rank (each re-sorts the tiles of ITS rows in every iteration; the layout is local
    and never changes a sum): the oracle's trajectory, also across a sharded checkpoint / resume"""
    po = oracle_built
    n, m, iters = 20011, 7, 12
    if mode == "fakerccl":
        monkeypatch.setenv("LBFGSB_RCCL_LIBRARY", _fake_rccl())
    monkeypatch.setenv("LBFGSB_TEST_COMPACT", "1")
    res = launch(world, mode, n, m, iters, variant, str(tmp_path / "out.json"))
    rows, x = oracle_rows(po, n, m, iters, variant is True)
    assert len(res["rows"]) == len(rows) == iters
    for a, b in zip(res["rows"], rows):
        assert a[:4] == b[:4], (a, b)
        assert a[4] == pytest.approx(b[4], rel=1e-9)
    assert np.max(np.abs(np.array(res["x"]) - x)) <= 1e-8 * max(1.0, np.max(np.abs(x)))

"""Option "compact_w": the two passes over W on the tile-local free-row layout (k_layout.hip) -- the
reference's cmprlb / subsm / formk loops run over Index(1:nfree) (src/lbfgsb.f90:1565-1583, :2743-2778, :1756-1793,
:2044-2054); streaming all n rows under a mask moves n / nfree times their bytes.

compact_w = 1 (what bench.py sets): the passes run on the layout while it is packed, the natural-order kernels before
the first pack (a problem whose rows are all free never pays for the option); compact_w = 2: always on the layout.
What must hold:
  * the LAYOUT never changes a result: with compact_w = 2 the sums are the same whatever the tiles look like --
    runs that never pack (policy 0), pack by the automatic rule (1) and re-sort the tiles in EVERY iteration (2)
    are bit for bit the same run (every return, x, g, the exported state);
  * against the run without the option only the order of the sums differs (another row-to-lane map): integer
    columns of the first iterations exactly, f to 1e-9, the oracle's trajectory as the production path follows it;
  * one-step parity against the oracle from imported states with the tiles re-sorted inside the step, every caller
    array incl. Ws / Wy after export (= packing + unpacking are permutations);
  * checkpoint / resume, sharded runs, the backtracking branch and dictionary-coded bounds with the option on.
The randomised differential test with the option on is in test_gpu_fuzz.py (switch compact_w=1).
"""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(oracle_built):
    import torch
    import lbfgsb_amd
    assert torch.cuda.is_available(), "these tests need the MI355X"
    lbfgsb_amd.load_library()
    return dict(po=oracle_built, torch=torch, la=lbfgsb_amd)


def _digest(t):
    return hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()


def _run(env, p, max_iter, pp=True, defer=False, export=True, **options):
    torch, la = env["torch"], env["la"]
    sol = la.DeviceSolver(p.n, p.m, defer_lnsrch=defer, same_stream_objective=defer, options=options or None)
    try:
        xs = [torch.from_numpy(p.x0.copy()).cuda(), torch.full((p.n,), 7.0, dtype=torch.float64, device="cuda")]
        gs = [torch.zeros_like(xs[0]), torch.full_like(xs[0], -3.0)]
        x, g = xs[0], gs[0]
        l, u = torch.from_numpy(p.l).cuda(), torch.from_numpy(p.u).cuda()
        nbd = torch.from_numpy(p.nbd.astype(np.int32)).cuda()
        rows, ints = [], []
        t = ""
        for _ in range(200000):
            if pp:
                t, cur = sol.setulb_pp(xs, l, u, nbd, gs, p.factr, p.pgtol)
                x, g = xs[cur], gs[cur]
            else:
                t = sol.setulb(x, l, u, nbd, g, p.factr, p.pgtol)
            if t.startswith("FG"):
                sol.sync()
                xh = x.cpu().numpy()
                gh = np.empty_like(xh)
                sol.f[0] = p.fg(xh, gh)
                g.copy_(torch.from_numpy(gh))
                torch.cuda.synchronize()
            else:
                sol.sync()
                rows.append((t, tuple(int(v) for v in sol.isave[21:44]), sol.f.tobytes(),
                             sol.dsave[[0, 1, 2, 3, 4, 10, 11, 12, 13, 14, 15]].tobytes(), _digest(x), _digest(g)))
                ints.append((int(sol.isave[29]), int(sol.isave[33]), int(sol.isave[32]), int(sol.isave[37]),
                             float(sol.f[0])))
                if not t.startswith("NEW_X") or sol.isave[29] >= max_iter:
                    break
        stats = sol.compact_stats()
        wa = iwa = None
        if export:
            wa, iwa = sol.export_state()
        return dict(rows=rows, ints=ints, stats=stats, task=t, wa=None if wa is None else wa.tobytes(),
                    iwa=None if iwa is None else iwa.tobytes(), x=x.cpu().numpy())
    finally:
        sol.close()


def _problems(po):
    from test_gpu_fuzz import make
    from test_gpu_parity import _random_box_rosenbrock
    ps = [po.problem_quadratic(20011, 7, mixed_nbd=True), po.problem_quadratic(4099, 10),
          po.problem_rosenbrock(1000, 10, 0.0, 0.0), po.problem_quadratic(300, 3, mixed_nbd=True),
          po.problem_quadratic(127, 5), po.problem_quadratic(129, 2, mixed_nbd=True)]
    ps += [make(po, s, 3000, 1, 11) for s in range(9100, 9112)]
    ps += [_random_box_rosenbrock(po, s) for s in (5003, 5021, 5006)]       # (reach subsm's backtracking branch)
    return [p for p in ps if p.m <= 10]


@pytest.mark.parametrize("pp,defer", [(True, False), (True, True), (False, False)])
def test_layout_never_changes_a_result(env, pp, defer):
    """policy 0 (never pack) == policy 1 (automatic) == policy 2 (re-sort every iteration), bit for bit"""
    po = env["po"]
    packed_runs = 0
    for p in _problems(po):
        base = _run(env, p, 60, pp=pp, defer=defer, compact_w=2, compact_policy=0)
        assert base["stats"][3] == 1 and base["stats"][0] == 0, (p.name, base["stats"])
        for pol in (2, 1):
            r = _run(env, p, 60, pp=pp, defer=defer, compact_w=2, compact_policy=pol, compact_min_rows=0)
            assert len(r["rows"]) == len(base["rows"]), (p.name, pol, len(r["rows"]), len(base["rows"]))
            for k, (a, b) in enumerate(zip(r["rows"], base["rows"])):
                assert a == b, "%s (n=%d m=%d) policy %d: return %d differs: %s | %s" % (p.name, p.n, p.m, pol, k,
                                                                                       a[:3], b[:3])
            assert r["wa"] == base["wa"] and r["iwa"] == base["iwa"], (p.name, pol, "exported state differs")
            if pol == 2:
                assert r["stats"][0] >= min(len(r["rows"]) - 2, 3), (p.name, r["stats"])   # it did re-sort
            packed_runs += r["stats"][0] > 0
    assert packed_runs >= 10


def test_option_on_follows_the_run_without_it(env):
    """another order of the sums, nothing else: integer columns of the first iterations, f to 1e-9"""
    po = env["po"]
    for p in _problems(po)[:8]:
        a = _run(env, p, 14, export=False)
        b = _run(env, p, 14, export=False, compact_w=2, compact_policy=2)
        c = _run(env, p, 14, export=False, compact_w=1, compact_min_rows=0)     # (what bench.py runs)
        assert a["stats"][3] == 0 and b["stats"][3] == 1 and c["stats"][3] == 1
        for rc in c["ints"][:8]:
            ra = a["ints"][rc[0] - 1]
            if rc[:4] != ra[:4]:
                break
            assert abs(ra[4] - rc[4]) <= 1e-9 * max(1.0, abs(ra[4])), (p.name, ra, rc)
        nsame = 0
        for ra, rb in zip(a["ints"], b["ints"]):
            if ra[:4] != rb[:4]:
                break
            assert abs(ra[4] - rb[4]) <= 1e-9 * max(1.0, abs(ra[4])), (p.name, ra, rb)
            nsame += 1
        assert nsame >= min(8, len(a["ints"])), (p.name, nsame, a["ints"][:nsame + 1], b["ints"][:nsame + 1])


CASES = [("quad1000", dict(kind="quad", n=1000, m=10), 70), ("quadmix4099", dict(kind="quadmix", n=4099, m=10), 64),
         ("rosenbrock1000", dict(kind="ros", n=1000, m=10, factr=0.0, pgtol=0.0), 60),
         ("quadmix777_m3", dict(kind="quadmix", n=777, m=3), 44), ("quad20011_m7", dict(kind="quad", n=20011, m=7), 40)]


@pytest.mark.parametrize("name,spec,ncalls", CASES, ids=[c[0] for c in CASES])
def test_two_step_parity_with_the_tiles_resorted_inside_the_step(env, name, spec, ncalls):
    """test_gpu_parity.py::test_production_path_two_step_parity with the option on and policy 2: call 1 runs the
    update pass on the layout (natural: just imported), call 2 re-sorts the tiles to the free set of that iteration,
    gathers the walk's records and patches WN1 through the layout, and its storing pass commits the pair INTO the
    layout; the export un-sorts.  Every caller array against the oracle's, integers exactly, floats to 1e-10."""
    from test_gpu_parity import _dev, compare_states, make_problem, oracle_snapshots
    po, torch, la = env["po"], env["torch"], env["la"]
    p = make_problem(po, spec)
    snaps = oracle_snapshots(po, p, ncalls)
    tested = packs = 0
    for k in range(len(snaps) - 2):
        s0, s1, s2 = snaps[k], snaps[k + 1], snaps[k + 2]
        if not (s0.task_s.startswith("FG_LN") and int(s0.isave[35]) == 1 and s1.task_s.startswith("NEW_X")
                and s2.task_s.startswith("FG_LN")):
            continue
        s = s0.copy()
        s.f[0] = p.fg(s.x, s.g)
        sol = la.DeviceSolver(p.n, p.m, options={"compact_w": 2, "compact_policy": 2})
        try:
            x, g = _dev(torch, s.x), _dev(torch, s.g)
            l, u, nbd = _dev(torch, p.l), _dev(torch, p.u), _dev(torch, p.nbd.astype(np.int32))
            sol.import_state(s.wa, s.iwa, s.isave)
            for name_ in ("task", "csave", "lsave", "isave", "dsave"):
                getattr(sol, name_)[:] = getattr(s, name_)
            sol.f[0] = s.f[0]

            def snapshot():
                torch.cuda.synchronize()
                wa, iwa = sol.export_state()
                return po.State(p.n, p.m, x.cpu().numpy(), g.cpu().numpy(), sol.f.copy(), wa, iwa,
                                sol.task.copy(), sol.csave.copy(), sol.lsave.copy(), sol.isave.copy(),
                                sol.dsave.copy())
            sol.setulb(x, l, u, nbd, g, p.factr, p.pgtol)
            out1 = snapshot()
            compare_states(out1, s1, p.n, p.m, po, skip=("xp",), check_indx2=False, check_iwhere=False)
            sol.setulb(x, l, u, nbd, g, p.factr, p.pgtol)
            packs += sol.compact_stats()[0]
            out2 = snapshot()
            compare_states(out2, s2, p.n, p.m, po, skip=("xp",), check_indx2=False)
        finally:
            sol.close()
        tested += 1
    assert tested >= 8 and packs >= tested - 2, (tested, packs)


def test_checkpoint_resume_with_the_option_on(env):
    """export (natural order) at iteration k, import into a fresh context with the option on, carry on: the rows
    of the uninterrupted run, bit for bit -- the layouts of the two runs differ, the sums do not"""
    po, torch, la = env["po"], env["torch"], env["la"]
    p = po.problem_quadratic(20011, 6, mixed_nbd=True)
    opts = {"compact_w": 2, "compact_policy": 1, "compact_min_rows": 0}

    def drive(sol, x, g, l, u, nbd, until, rows):
        while True:
            t = sol.setulb(x, l, u, nbd, g, 0.0, 0.0)
            if t.startswith("FG"):
                sol.f[0] = sol.objective(0, x, g)
            elif t.startswith("NEW_X"):
                rows.append((int(sol.isave[29]), int(sol.isave[33]), int(sol.isave[32]), int(sol.isave[37]),
                             sol.f.tobytes(), _digest(x)))
                if sol.isave[29] >= until:
                    return
            else:
                return

    def tensors():
        return (torch.from_numpy(p.x0.copy()).cuda(), torch.zeros(p.n, dtype=torch.float64, device="cuda"),
                torch.from_numpy(p.l).cuda(), torch.from_numpy(p.u).cuda(),
                torch.from_numpy(p.nbd.astype(np.int32)).cuda())
    ref = []
    sol = la.DeviceSolver(p.n, p.m, options=opts)
    x, g, l, u, nbd = tensors()
    drive(sol, x, g, l, u, nbd, 16, ref)
    assert sol.compact_stats()[0] >= 1      # the automatic rule did pack (a third of the rows sits at a bound)
    sol.close()
    for cut in (3, 8, 11):
        rows = []
        a = la.DeviceSolver(p.n, p.m, options=opts)
        x, g, l, u, nbd = tensors()
        drive(a, x, g, l, u, nbd, cut, rows)
        torch.cuda.synchronize()
        wa, iwa = a.export_state()
        b = la.DeviceSolver(p.n, p.m, options=opts)
        b.import_state(wa, iwa, a.isave)
        for name in ("task", "csave", "lsave", "isave", "dsave", "f"):
            getattr(b, name)[:] = getattr(a, name)
        a.close()
        drive(b, x, g, l, u, nbd, 16, rows)
        b.close()
        assert rows == ref, (cut, rows[:3], ref[:3])


def test_full_size_rows_with_the_option_on(env):
    """n = 1e6, m = 10, uniform bounds, ping-pong + DEFER (the entry bench.py times): SURVEY.md 8c's anchors of this
    problem -- integer columns exactly, f to 1e-9 -- and the layout packed by the automatic rule"""
    torch, la = env["torch"], env["la"]
    n, m = 1_000_000, 10
    sol = la.DeviceSolver(n, m, defer_lnsrch=True, same_stream_objective=True,
                          options={"compact_w": 1, "compact_min_rows": 0})
    try:
        xs = [torch.zeros(n, dtype=torch.float64, device="cuda"), torch.empty(n, dtype=torch.float64, device="cuda")]
        gs = [torch.zeros_like(xs[0]), torch.empty_like(xs[0])]
        l, u = torch.full_like(xs[0], -1.0), torch.full_like(xs[0], 1.0)
        nbd = torch.full((n,), 2, dtype=torch.int32, device="cuda")
        rows = {}
        while True:
            t, cur = sol.setulb_pp(xs, l, u, nbd, gs, 0.0, 0.0)
            if t.startswith("FG"):
                sol.objective(0, xs[cur], gs[cur], deferred=True)
            elif t.startswith("NEW_X"):
                it = int(sol.isave[29])
                rows[it] = (int(sol.isave[33]), int(sol.isave[32]), int(sol.isave[37]), float(sol.f[0]))
                if it >= 30:
                    break
            else:
                raise AssertionError(t)
        packs, unpacks, packed, elig = sol.compact_stats()
    finally:
        sol.close()
    assert elig == 1 and packs >= 1 and packed == 1 and unpacks == 0, (packs, unpacks, packed, elig)
    assert rows[1][1:3] == (976721, 23280) and abs(rows[1][3] - 8.2541454907951783e06) <= 1e-9 * 8.25e6
    assert rows[2][2] == 499997 and abs(rows[2][3] - 4.4408007558922265e06) <= 1e-9 * 4.44e6
    assert abs(rows[3][3] - 4.2654335281014517e06) <= 1e-9 * 4.27e6
    assert rows[10][1:3] == (39, 499959) and abs(rows[10][3] - 4.2089636688019084e06) <= 1e-9 * 4.21e6
    assert abs(rows[20][3] - 4.2086430506394058e06) <= 1e-9 * 4.21e6
    assert rows[30][0] == 32 and abs(rows[30][3] - 4.2086404848337891e06) <= 1e-9 * 4.21e6

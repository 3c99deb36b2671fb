"""LBFGSB_F_DEFER_LNSRCH (include/lbfgsb_hip.h): the storing pass's line-search sums travel with the
NEXT call's fetch instead of being waited for -- one host round trip per iteration less.  The flag
must change nothing a caller can observe at a 'NEW_X' return: same arithmetic in the same order.
These tests run every problem twice -- a default context and a deferring one -- and require every
NEW_X return, the final return and the exported state to be BIT FOR BIT equal; the only visible
difference allowed is the re-issued 'FG_LNSRCH' request in the calls where the set-up asks for
another point than x = z (subsm's backtracking step, an ascent direction), which the library
counts (lbfgsb_hip_defer_stats) and the harness checks against its own count of FG requests.
The default mode itself is pinned against the oracle by test_gpu_fuzz.py / test_gpu_parity.py."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _digest(t):
    return hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()


def _run(p, pp, defer, max_iter, **ctx):
    """drive p through one context; f, g are evaluated on the host from x as it is on the solver's
    stream (sol.sync() first: a deferring context does not synchronise at an FG return)"""
    import torch
    import lbfgsb_amd as la
    # (a deferring context must declare that its objective is ordered on the solver's stream: this harness
    #  synchronises the solver before it reads x and the device after it has written g)
    sol = la.DeviceSolver(p.n, p.m, defer_lnsrch=defer, same_stream_objective=defer, **ctx)
    try:
        xs = [torch.from_numpy(p.x0.copy()).cuda(), torch.full((p.n,), 7.0, dtype=torch.float64, device="cuda")]
        gs = [torch.zeros_like(xs[0]), torch.full_like(xs[0], -3.0)]
        x, g = xs[0], gs[0]
        l, u = torch.from_numpy(p.l).cuda(), torch.from_numpy(p.u).cuda()
        nbd = torch.from_numpy(p.nbd.astype(np.int32)).cuda()
        rows, nfg_req = [], 0
        t = ""
        for _ in range(200000):
            if pp:
                t, cur = sol.setulb_pp(xs, l, u, nbd, gs, p.factr, p.pgtol)
                x, g = xs[cur], gs[cur]
            else:
                t = sol.setulb(x, l, u, nbd, g, p.factr, p.pgtol)
            if t.startswith("FG"):
                nfg_req += 1
                sol.sync()
                xh = x.cpu().numpy()
                gh = np.empty_like(xh)
                sol.f[0] = p.fg(xh, gh)
                g.copy_(torch.from_numpy(gh))
                torch.cuda.synchronize()
            else:
                sol.sync()
                rows.append((t, tuple(int(v) for v in sol.isave[21:44]), sol.f.tobytes(),
                             sol.dsave[[0, 1, 2, 3, 4, 10, 11, 12, 13, 14, 15]].tobytes(),   # (5..9 are wall-clock timers)
                             _digest(x), _digest(g)))
                if not t.startswith("NEW_X") or sol.isave[29] >= max_iter:
                    break
        wa, iwa = sol.export_state()
        return dict(rows=rows, nfg_req=nfg_req, defer=sol.defer_stats(), wa=wa.tobytes(), iwa=iwa.tobytes(),
                    task=t, nfgv=int(sol.isave[33]))
    finally:
        sol.close()


def _same(p, pp, max_iter=60, **ctx):
    a = _run(p, pp, False, max_iter, **ctx)
    b = _run(p, pp, True, max_iter, **ctx)
    assert a["defer"] == (0, 0)
    assert len(a["rows"]) == len(b["rows"]), (p.name, len(a["rows"]), len(b["rows"]), a["task"], b["task"])
    for k, (ra, rb) in enumerate(zip(a["rows"], b["rows"])):
        assert ra == rb, "%s (n=%d m=%d pp=%s): return %d differs: %s | %s" % (p.name, p.n, p.m, pp, k, ra[:3], rb[:3])
    assert a["wa"] == b["wa"] and a["iwa"] == b["iwa"], (p.name, "exported state differs")
    deferred, reissued = b["defer"]
    # every request the default mode made, plus the ones the deferring mode had to make twice
    assert b["nfg_req"] == a["nfg_req"] + reissued, (p.name, a["nfg_req"], b["nfg_req"], b["defer"])
    assert a["nfgv"] == b["nfgv"]
    return deferred, reissued


@pytest.mark.parametrize("pp", [True, False])
def test_random_problems_bit_identical(oracle_built, pp):
    from test_gpu_fuzz import make
    po = oracle_built
    tot = [0, 0]
    for seed in list(range(200, 260)) + list(range(5200, 5212)):
        p = make(po, seed, 400, 1, 13) if seed < 5000 else make(po, seed, 3000, 11, 33)
        d, r = _same(p, pp)
        tot[0] += d
        tot[1] += r
    assert tot[0] > 500, tot     # the flag did act: most iterations deferred their set-up


@pytest.mark.parametrize("pp", [True, False])
def test_wide_memories_bit_identical(oracle_built, pp):
    """m > 32: the r pass as one launch over all columns defers its four sums like subsm_update_kernel does for
    m <= 32 (they ride with the fetch of the split update pass one call later); an uphill projected step lands in
    the unfused steps.  Every return equals the run that waits for them in the same call."""
    from test_gpu_fuzz import make
    po = oracle_built
    tot = [0, 0]
    for seed in range(8800, 8830):
        p = make(po, seed, 1200, 33, 80)
        d, r = _same(p, pp, max_iter=100)
        tot[0] += d
        tot[1] += r
    assert tot[0] > 800, tot     # (deferred also while col > 32)


@pytest.mark.parametrize("pp", [True, False])
def test_backtracking_and_restarts_bit_identical(oracle_built, pp):
    """problems that reach subsm's backtracking branch (:2830-2879) and the 'refresh the memory' branches:
    the deferred set-up has to go back to the iterate, put iwhere back as the walk left it, run the
    branch and re-issue its request"""
    from test_gpu_parity import _random_box_rosenbrock
    po = oracle_built
    reissued = 0
    for seed in [5003, 5021, 5028, 5006, 5007, 5040, 5041, 5042, 5043, 5044, 5045, 5046, 5047, 5048, 5049]:
        p = _random_box_rosenbrock(po, seed)
        _, r = _same(p, pp, max_iter=150)
        reissued += r
    assert reissued > 0, "these seeds are expected to re-issue at least one request"


@pytest.mark.parametrize("family", ["linear", "lattice", "tiny", "rosenchain", "scaled"])
def test_families_bit_identical(oracle_built, family):
    import test_gpu_fuzz as tf
    po = oracle_built
    gen = getattr(tf, "fam_" + family)
    for seed in range(61000, 61012):
        _same(gen(po, seed), True)
        _same(gen(po, seed), False)


def test_export_is_refused_while_deferred_and_start_resets(oracle_built):
    import torch
    import lbfgsb_amd as la
    from test_gpu_fuzz import make
    po = oracle_built
    p = make(po, 207, 400, 4, 9)
    with pytest.raises(ValueError, match="same_stream_objective"):   # the flag skips the sync at FG_LNSRCH returns
        la.DeviceSolver(p.n, p.m, defer_lnsrch=True)
    plain = la.DeviceSolver(p.n, p.m)
    with pytest.raises(ValueError, match="same_stream_objective"):
        plain.set_option("defer_lnsrch", 1)
    plain.close()
    sol = la.DeviceSolver(p.n, p.m, defer_lnsrch=True, same_stream_objective=True)
    x = torch.from_numpy(p.x0.copy()).cuda()
    g = torch.zeros_like(x)
    l, u = torch.from_numpy(p.l).cuda(), torch.from_numpy(p.u).cuda()
    nbd = torch.from_numpy(p.nbd.astype(np.int32)).cuda()

    def step():
        t = sol.setulb(x, l, u, nbd, g, p.factr, p.pgtol)
        if t.startswith("FG"):
            sol.sync()
            xh = x.cpu().numpy()
            gh = np.empty_like(xh)
            sol.f[0] = p.fg(xh, gh)
            g.copy_(torch.from_numpy(gh))
            torch.cuda.synchronize()
        return t
    refused = False
    for _ in range(400):
        before = sol.defer_stats()[0]
        t = step()
        if t.startswith("FG_LN") and sol.defer_stats()[0] > before:
            with pytest.raises(la.LbfgsbError):
                sol.export_state()
            refused = True
            break
        if not (t.startswith("FG") or t.startswith("NEW_X")):
            break
    assert refused, "no deferred set-up seen"
    # START in the middle of a deferred set-up: the new run is the run of a fresh context
    sol.task[:] = la.solver.pad60("START")
    x.copy_(torch.from_numpy(p.x0))
    rows = []
    for _ in range(60):
        t = step()
        if t.startswith("NEW_X"):
            rows.append((int(sol.isave[29]), int(sol.isave[33]), sol.f.tobytes(), _digest(x)))
        elif not t.startswith("FG"):
            break
    sol.close()
    ref = _run(p, False, True, 10 ** 6)
    got = [(r[1][8], r[1][12], r[2], r[4]) for r in ref["rows"] if r[0].startswith("NEW_X")][:len(rows)]
    assert rows == got


@pytest.mark.parametrize("pp,defer", [(True, False), (False, False), (True, True)])
def test_launch_plumbing_options_change_nothing(oracle_built, pp, defer):
    """The round-4 changes to HOW results travel -- the finalize that publishes by itself (`spin`), parked
    finalize jobs (`fold_finalize`), formk's patch queued behind freev's counting pass (`eager_patch`), the same
    chain queued speculatively behind the evaluation of a trial point (`spec_freev`) -- must not change a single
    bit of WHAT is computed: every return of a run with all four switched off (the
    round-3 plumbing: one finalize per kernel, a D2H copy + stream sync per phase, the patch after a host round
    trip of its own) equals the default's, on random problems whose free sets change from iteration to
    iteration (so that the patch really runs)."""
    from test_gpu_fuzz import make, fam_rosenchain
    po = oracle_built
    old = {"spin": 0, "fold_finalize": 0, "eager_patch": 0, "spec_freev": 0}
    new = {"spec_freev": 1}     # (opt-in; everything else is on by default)
    for seed in list(range(700, 724)) + list(range(5300, 5306)):
        p = make(po, seed, 400, 1, 13) if seed < 5000 else make(po, seed, 3000, 11, 33)
        a = _run(p, pp, defer, 60, options=old)
        b = _run(p, pp, defer, 60, options=new)
        assert a["rows"] == b["rows"] and a["wa"] == b["wa"] and a["iwa"] == b["iwa"], (p.name, p.n, p.m)
    for seed in range(62000, 62008):
        p = fam_rosenchain(po, seed)
        a = _run(p, pp, defer, 80, options=old)
        b = _run(p, pp, defer, 80, options=new)
        assert a["rows"] == b["rows"] and a["wa"] == b["wa"], (p.name, p.n, p.m)


@pytest.mark.parametrize("pp", [False, True], ids=["classic", "pingpong"])
def test_the_reach_of_the_first_window_changes_nothing(oracle_built, pp):
    """The first window of a breakpoint walk asks further ahead than the walk needs when it starts (option
    `win_slack`, default 0.25), so that the stationary point's drift while breakpoints are crossed does not cost a
    second pass over x, g.  More candidates ride back, the walk consumes the same ones in the same order: every return
    of a run equals the run with windows that end exactly at the first estimate, and the run whose windows reach
    three times as far (lists beyond the fast path's 256 candidates: the sorted route)."""
    from test_gpu_fuzz import make, fam_rosenchain
    po = oracle_built
    for seed in list(range(900, 916)) + list(range(5400, 5404)):
        p = make(po, seed, 600, 1, 13) if seed < 5000 else make(po, seed, 3000, 11, 33)
        a = _run(p, pp, True, 60, options={"win_slack": 0})
        for slack in (0.25, 2.0):
            b = _run(p, pp, True, 60, options={"win_slack": slack})
            assert a["rows"] == b["rows"] and a["wa"] == b["wa"] and a["iwa"] == b["iwa"], (p.name, p.n, p.m, slack)
    for seed in range(62100, 62104):
        p = fam_rosenchain(po, seed)
        a = _run(p, pp, True, 80, options={"win_slack": 0})
        b = _run(p, pp, True, 80, options={"win_slack": 0.25})
        assert a["rows"] == b["rows"] and a["wa"] == b["wa"], (p.name, p.n, p.m)


def _run_torch_objective(p_data, pp, ordered, max_iter, side_stream):
    """a separable quadratic whose f, g are evaluated by torch ops on torch's CURRENT stream (optionally a side
    stream).  ordered=False: a default context (host sync at every FG return), f read back with .item();
    ordered=True: stream_ordered + defer_lnsrch -- events in both directions (lbfgsb_hip_return_event /
    lbfgsb_hip_wait_stream), f handed over as a device scalar (lbfgsb_hip_f_device): no host sync in the caller."""
    import torch
    import lbfgsb_amd as la
    n, m, a_h, c_h, l_h, u_h, nbd_h, x0_h = p_data
    sol = la.DeviceSolver(n, m, stream_ordered=ordered, defer_lnsrch=ordered)
    st = torch.cuda.Stream() if side_stream else torch.cuda.current_stream()
    try:
        with torch.cuda.stream(st):
            a, c = torch.from_numpy(a_h).cuda(), torch.from_numpy(c_h).cuda()
            xs = [torch.from_numpy(x0_h.copy()).cuda(), torch.full((n,), 7.0, dtype=torch.float64, device="cuda")]
            gs = [torch.zeros_like(xs[0]), torch.full_like(xs[0], -3.0)]
            l, u = torch.from_numpy(l_h).cuda(), torch.from_numpy(u_h).cuda()
            nbd = torch.from_numpy(nbd_h).cuda()
            x, g = xs[0], gs[0]
            rows, nfg_req = [], 0
            t = ""
            for _ in range(100000):
                if pp:
                    t, cur = sol.setulb_pp(xs, l, u, nbd, gs, 0.0, 0.0)
                    x, g = xs[cur], gs[cur]
                else:
                    t = sol.setulb(x, l, u, nbd, g, 0.0, 0.0)
                if t.startswith("FG"):
                    nfg_req += 1
                    d = x - c
                    torch.mul(a, d, out=g)
                    f = 0.5 * torch.dot(g, d)
                    if ordered:
                        sol.set_f_device(f)
                    else:
                        sol.f[0] = float(f.item())
                else:
                    sol.sync()
                    rows.append((t, tuple(int(v) for v in sol.isave[21:44]), sol.f.tobytes(),
                                 sol.dsave[[0, 1, 2, 3, 4, 10, 11, 12, 13, 14, 15]].tobytes(), _digest(x), _digest(g)))
                    if not t.startswith("NEW_X") or sol.isave[29] >= max_iter:
                        break
            st.synchronize()
        return dict(rows=rows, nfg_req=nfg_req, defer=sol.defer_stats(), stats=sol.stats())
    finally:
        sol.close()


@pytest.mark.parametrize("pp", [True, False])
@pytest.mark.parametrize("side_stream", [False, True])
def test_torch_stream_objective_may_defer(oracle_built, pp, side_stream):
    """An ordinary stream-ordered caller (objective = torch ops on torch's stream) with LBFGSB_F_DEFER_LNSRCH: events
    instead of host syncs in both directions, f as a device scalar.  Every NEW_X return bit for bit that of the
    default context driven by the same objective with f read back on the host; the library's own host syncs per
    iteration drop by one, the caller adds none."""
    rng = np.random.default_rng(42)
    for n, m in ((20011, 7), (100003, 10)):
        a = 1.0 + 99.0 * rng.random(n)
        c = rng.normal(0, 2, n)
        l = np.full(n, -1.0) - 1e-3 * rng.random(n)
        u = np.full(n, 1.0) + 1e-3 * rng.random(n)
        nbd = rng.integers(0, 4, n).astype(np.int32)
        x0 = np.zeros(n)
        data = (n, m, a, c, l, u, nbd, x0)
        base = _run_torch_objective(data, pp, False, 40, side_stream)
        got = _run_torch_objective(data, pp, True, 40, side_stream)
        assert len(base["rows"]) == len(got["rows"]) >= 30
        for k, (ra, rb) in enumerate(zip(base["rows"], got["rows"])):
            assert ra == rb, (n, m, pp, side_stream, k, ra[:3], rb[:3])
        deferred, reissued = got["defer"]
        assert deferred >= 25 and got["nfg_req"] == base["nfg_req"] + reissued
        iters = len(got["rows"])
        assert got["stats"]["syncs"] <= base["stats"]["syncs"] - 0.8 * iters, (got["stats"]["syncs"],
                                                                             base["stats"]["syncs"], iters)


@pytest.mark.parametrize("pp", [True, False])
def test_searches_with_three_or_more_trials_bit_identical(oracle_built, pp):
    """ problems whose line searches take 3 - 20 trial points, m <= 32 and m > 32: deferring the set-up
    changes no return, although the second trial is evaluated by the update pass and then rejected"""
    from test_gpu_fuzz import MULTI_TRIAL_SMALL, MULTI_TRIAL_WIDE, make
    po = oracle_built
    for seed, lo, hi in MULTI_TRIAL_SMALL + MULTI_TRIAL_WIDE:
        _same(make(po, seed, 1200, lo, hi), pp, max_iter=60)

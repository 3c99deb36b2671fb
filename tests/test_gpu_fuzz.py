"""Randomised differential test: random box-constrained problems (random n, m, bound types incl.
fixed and unbounded variables, separable + coupled + non-convex objectives, random factr/pgtol)
solved call by call through the device-pointer entry and by the oracle.  Every setulb return
must match (task, iteration, nfg, nseg, nfree; f to 1e-8); where a run leaves the oracle's
trajectory, the split must be reproduced by ONE oracle call from the GPU's own previous state
(one-step parity: integers exactly, floats to 1e-10), and the final f agrees to 1e-7.  There is no
allowance for unexplained splits.  Exercises the production iteration (speculative update pass,
pending pair, functional Cauchy point, skipped updates, restarts) on shapes no hand-written case
covers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make(po, seed, nmax, mlo, mhi):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, nmax))
    m = int(rng.integers(mlo, mhi))
    a = 1.0 + 99.0 * rng.random(n)
    c = rng.normal(0, 2, n)
    mu = float(rng.choice([0.0, 0.5, 5.0]))
    kind = int(rng.integers(0, 3))

    def fg(x, g):
        d = x - c
        f = 0.5 * np.sum(a * d * d)
        g[:] = a * d
        if n > 1 and mu > 0:
            e = x[:-1] - x[1:]
            f += 0.5 * mu * np.sum(e * e)
            g[:-1] += mu * e
            g[1:] -= mu * e
        if kind == 2:       # non-convex term: provokes several line-search trials
            f += np.sum(np.cos(3 * x))
            g[:] -= 3 * np.sin(3 * x)
        return float(f)
    l = rng.normal(-1, 1, n)
    u = l + np.abs(rng.normal(1.5, 1, n))
    fixed = rng.random(n) < 0.03
    u[fixed] = l[fixed]
    nbd = rng.integers(0, 4, n).astype(np.int32)
    if rng.random() < 0.15:
        nbd[:] = 0
    if rng.random() < 0.15:
        nbd[:] = 2
    x0 = rng.normal(0, 3, n)
    factr = 0.0 if rng.random() < 0.5 else 1e7
    pgtol = 0.0 if rng.random() < 0.5 else 1e-5
    return po.Problem("fuzz%d" % seed, n, m, x0, l, u, nbd, factr, pgtol, fg, np.float64)


def _state(po, p, sol, x, g):
    """the caller arrays of a DEFAULT (production-path) context after a setulb return; export_state is
    read-only: it writes z and d out where the lean subspace pass left them implicit, nothing else"""
    import torch
    torch.cuda.synchronize()
    wa, iwa = sol.export_state()
    return po.State(p.n, p.m, x.cpu().numpy(), g.cpu().numpy(), sol.f.copy(), wa, iwa, sol.task.copy(),
                    sol.csave.copy(), sol.lsave.copy(), sol.isave.copy(), sol.dsave.copy())


def _explain_divergence(po, p, prev, got, pp=False):
    """The trajectories parted ways at this call.  One ORACLE call from the GPU's own previous
    state (every caller array as the production context exported it, f and g as the test evaluated
    them at the GPU's x) must reproduce the GPU's call: task, every counter, iwhere exactly, floats to
    1e-10 -- then the GPU took a correct reference step from a state that differed from the oracle's
    trajectory by rounding, and the split is drift, not a defect."""
    from test_gpu_parity import compare_states
    s = prev.copy()
    po.call(po.Engine("oracle"), p, s)
    if pp and got.task_s.startswith("FG"):
        # ping-pong entry: at an 'FG' return g[cur] is the buffer the caller is about to write into,
        # not the old gradient (that one lives on as r, which the comparison below covers)
        got = got.copy()
        got.g = s.g.copy()
    # by design: at a NEW_X return a production context already holds the iwhere
    # pattern of the NEXT cauchy scan; xp / the enter-leave half of Indx2 are not materialised
    # ... and after a REJECTED first trial (FG_LNSRCH in, FG_LNSRCH out) the pattern of the rejected point
    compare_states(got, s, p.n, p.m, po, skip=("xp",), check_lists=False,
                   check_iwhere=got.task_s.startswith("FG_LN") and not prev.task_s.startswith("FG_LN"),
                   stpmx_cond=True)


def drive_with_replay(po, p, max_iter, pp=False, **ctx):
    """Run p on the GPU (default context + ctx), call by call beside the oracle's trajectory.
    -> (split, n_calls): split = index of the first call that differs from the oracle's trajectory
    (None: equal to the end).  A split that one oracle call from the GPU's previous state does not
    reproduce raises.  pp: through the ping-pong entry (lbfgsb_hip_setulb_dev_pp)."""
    import torch
    import lbfgsb_amd as la

    def row(t, isave, f):
        return (t[:12], int(isave[29]), int(isave[33]), int(isave[32]), int(isave[37]), float(f))
    ro = []
    so = po.run(po.Engine("oracle"), p, max_iter=max_iter,
                snapshot=lambda k, s: ro.append(row(s.task_s, s.isave, s.f[0])))
    sol = la.DeviceSolver(p.n, p.m, **ctx)
    try:
        xs = [torch.from_numpy(p.x0.copy()).cuda(), torch.full((p.n,), 7.0, dtype=torch.float64, device="cuda")]
        gs = [torch.zeros_like(xs[0]), torch.full_like(xs[0], -3.0)]
        x, g = xs[0], gs[0]
        l, u = torch.from_numpy(p.l).cuda(), torch.from_numpy(p.u).cuda()
        nbd = torch.from_numpy(p.nbd.astype(np.int32)).cuda()
        rg, prev, split = [], None, None
        for _ in range(100000):
            if pp:
                t, cur_i = sol.setulb_pp(xs, l, u, nbd, gs, p.factr, p.pgtol)
                x, g = xs[cur_i], gs[cur_i]
            else:
                t = sol.setulb(x, l, u, nbd, g, p.factr, p.pgtol)
            rg.append(row(t, sol.isave, sol.f[0]))
            k = len(rg) - 1
            cur = None
            if split is None:
                cur = _state(po, p, sol, x, g)
                same = (k < len(ro) and ro[k][:5] == rg[k][:5]
                        and abs(ro[k][5] - rg[k][5]) <= 1e-8 * max(1.0, abs(ro[k][5])))
                if not same:
                    split = k
                    assert prev is not None, (p.name, "diverged at the very first call", ro[:1], rg[:1])
                    try:
                        _explain_divergence(po, p, prev, cur, pp)
                    except AssertionError as e:
                        raise AssertionError(
                            "%s (n=%d m=%d): call %d differs from the oracle's trajectory (%s vs %s) and is "
                            "NOT reproduced by one oracle call from the GPU's previous state: %s"
                            % (p.name, p.n, p.m, k, ro[k:k + 1], rg[k:k + 1], e))
            if t.startswith("FG"):
                xh = x.cpu().numpy()
                gh = np.empty_like(xh)
                sol.f[0] = p.fg(xh, gh)
                g.copy_(torch.from_numpy(gh))
                if cur is not None:
                    cur.f[0], cur.g = sol.f[0], gh.copy()
            elif t.startswith("NEW_X"):
                if sol.isave[29] >= max_iter:
                    break
            else:
                break
            prev = cur
        fgp, fo = float(sol.f[0]), float(so.f[0])
    finally:
        sol.close()
    if split is None:
        assert len(rg) == len(ro), (p.name, len(rg), len(ro))
    else:
        assert abs(fo - fgp) <= 1e-7 * max(1.0, abs(fo)), (p.name, p.n, p.m, split, len(ro), len(rg), fo, fgp)
    return split, len(ro)


@pytest.mark.parametrize("first,count,nmax,mlo,mhi,switch", [
    (0, 120, 400, 1, 13, None), (5000, 40, 3000, 11, 33, None),
    # the ping-pong entry (t = x, r = g as a change of roles between two caller buffer pairs)
    (0, 120, 400, 1, 13, "pp"), (5000, 40, 3000, 11, 33, "pp"), (7300, 40, 1500, 1, 25, "pp,lean=0"),
    # m > 32: the iteration out of unfused tile primitives (solver_wide.inl)
    (8000, 30, 1200, 33, 80, None), (8100, 20, 1200, 33, 80, "pp"),
    # the measurement switches select fallback paths that must stay correct: the candidate
    # hand-over of the update pass, and the three-pass iteration (no closed form, stored z and d)
    (7000, 40, 1500, 1, 25, "spec_capture=1"), (7100, 40, 1500, 1, 25, "two_pass=0"),
    (7200, 40, 1500, 1, 25, "lean=0")])
def test_random_problems_against_oracle(oracle_built, first, count, nmax, mlo, mhi, switch):
    po = oracle_built
    words = switch.split(",") if switch else []
    opts = {w.split("=")[0]: float(w.split("=")[1]) for w in words if "=" in w}
    splits = []
    for seed in range(first, first + count):
        split, ncalls = drive_with_replay(po, make(po, seed, nmax, mlo, mhi), 80, pp="pp" in words, options=opts)
        if split is not None:
            splits.append((seed, split, ncalls))
    print("%d of %d runs left the oracle's trajectory, each reproduced by a one-step oracle replay "
          "(seed, first differing call, calls): %s" % (len(splits), count, splits))


def test_random_problems_parallel_gcp_search(oracle_built):
    """The opt-in parallel GCP search on random problems: option pg_min = 0 sends EVERY walk that
    passes its first breakpoint through it -- the closed form when no pair is stored, the sort +
    scans (with the f2 clamp) when pairs are stored -- on all bound types, fixed and unbounded
    variables, m = 1..12.  Against the oracle's sequential walk: same iteration / nfg columns,
    nseg and nfree within 2, f to 1e-8, call by call over the first 12
    iterations; a run may part ways late only at rounding level (final f to 1e-7).  (The flag is a
    declared deviation from the reference's arithmetic: a one-step replay against the sequential
    oracle is not the bar here.)"""
    po = oracle_built
    import torch
    import lbfgsb_amd as la
    searched, late, count = 0, 0, 60
    for seed in range(9000, 9000 + count):
        p = make(po, seed, 2500, 1, 13)
        if not np.any(p.nbd != 0):
            continue
        ro = []
        so = po.run(po.Engine("oracle"), p, max_iter=12,
                    snapshot=lambda k, s: ro.append((int(s.isave[29]), int(s.isave[33]), int(s.isave[32]),
                                                     int(s.isave[37]), float(s.f[0])))
                    if s.task_s.startswith("NEW_X") else None)
        sol = la.DeviceSolver(p.n, p.m, parallel_gcp=True, options={"pg_min": 0})
        x = torch.from_numpy(p.x0.copy()).cuda()
        g = torch.zeros_like(x)
        l, u = torch.from_numpy(p.l).cuda(), torch.from_numpy(p.u).cuda()
        nbd = torch.from_numpy(p.nbd.astype(np.int32)).cuda()
        rg = []
        for _ in range(100000):
            t = sol.setulb(x, l, u, nbd, g, p.factr, p.pgtol)
            if t.startswith("FG"):
                xh = x.cpu().numpy()
                gh = np.empty_like(xh)
                sol.f[0] = p.fg(xh, gh)
                g.copy_(torch.from_numpy(gh))
            elif t.startswith("NEW_X"):
                rg.append((int(sol.isave[29]), int(sol.isave[33]), int(sol.isave[32]), int(sol.isave[37]),
                           float(sol.f[0])))
                if sol.isave[29] >= 12:
                    break
            else:
                break
        searched += sol.stats()["cauchy_fullsorts"]
        fgp = float(sol.f[0])
        sol.close()
        k = 0
        while (k < min(len(ro), len(rg)) and ro[k][:2] == rg[k][:2] and abs(ro[k][2] - rg[k][2]) <= 2
               and abs(ro[k][3] - rg[k][3]) <= 2 and abs(ro[k][4] - rg[k][4]) <= 1e-8 * max(1.0, abs(ro[k][4]))):
            k += 1
        if k == len(ro) == len(rg):
            continue
        late += 1
        fo = float(so.f[0])
        assert k >= 0.4 * len(ro) and abs(fo - fgp) <= 1e-7 * max(1.0, abs(fo)), \
            (seed, p.n, p.m, k, len(ro), len(rg), ro[max(0, k - 1):k + 1], rg[max(0, k - 1):k + 1])
    assert searched >= 60           # the sort + scan path really ran (full sorts are its signature)
    assert late <= 0.2 * count


def test_random_problems_exact_tie_order(oracle_built):
    """Option exact_always = 1: EVERY walk of 40 random problems is replayed in the order of the
    reference's heap (all breakpoint times on the host, hpsolb, records gathered in pop order)
    instead of the device's (t, index) order -- the route a walk that ends inside a group of equal
    breakpoints takes by default.  Same bar as the main differential test: every return equals the
    oracle's; a split must be reproduced by a one-step oracle replay."""
    po = oracle_built
    splits = []
    for seed in range(9500, 9540):
        split, ncalls = drive_with_replay(po, make(po, seed, 1500, 1, 13), 30, options={"exact_always": 1})
        if split is not None:
            splits.append((seed, split, ncalls))
    print("exact order: %d of 40 runs left the oracle's trajectory, each reproduced one-step: %s"
          % (len(splits), splits))

"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.

Bars (BASELINE.json north_star): bit-exact for active-set / index / counter work;
<= 1e-10 relative on fp64 f, g and the n-vectors at IDENTICAL iterates (one call made
from the same caller state on both sides).  Whole-trajectory tests use integer columns
exactly and floats with a drift allowance, because reductions are summed in a different
order on the GPU (cross-compiler drift is already ~1e-11 after 23 iterations).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TIME_D = [5, 6, 7, 8, 9]      # dsave(6:10): wall-clock slots
RTOL = 1e-10


@pytest.fixture(scope="module")
def env(oracle_built):
    import torch
    import lbfgsb_amd
    assert torch.cuda.is_available(), "these tests need the MI355X"
    lbfgsb_amd.load_library()
    return dict(po=oracle_built, torch=torch, la=lbfgsb_amd)


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def nrm_close(a, b, rtol, what, floor=0.0):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if b.size and a.shape == b.shape and not (np.isfinite(a).all() and np.isfinite(b).all()):
        # non-finite values -- the reference's own arithmetic on degenerate problems (a linear objective:
        # 0/0 in the line search once d = 0): the same kind in the same places, the rest as usual
        assert np.array_equal(np.isnan(a), np.isnan(b)), what + ": NaN in different places"
        inf = np.isinf(a) | np.isinf(b)
        assert np.array_equal(a[inf], b[inf]), what + ": infinities differ"
        keep = np.isfinite(a) & np.isfinite(b)
        a, b = a[keep], b[keep]
    scale = max(float(np.max(np.abs(b))) if b.size else 0.0, floor)
    err = float(np.max(np.abs(a - b))) if b.size else 0.0
    assert err <= rtol * scale + 1e-300, "%s: max|diff| %.3e > %.1e * %.3e" % (what, err, rtol, scale)


def rel_err(a, b, floor=0.0):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    scale = max(float(np.max(np.abs(b))) if b.size else 0.0, floor, 1e-300)
    return (float(np.max(np.abs(a - b))) if b.size else 0.0) / scale


# the largest amplification seen: error of a factor / (condition number x error of what was factored)
COND_SEEN = {"wt": 0.0, "wn": 0.0}


def compare_states(got, exp, n, m, po, rtol=RTOL, check_lists=True, skip=(), check_indx2=True,
                   check_iwhere=True, stpmx_cond=False):
    """got / exp: pyoracle.State after the same call from the same input state.
    stpmx_cond (states near convergence, where d = z - x is a few ulps of x): stpmx = (bound - x_i) /
    d_i inherits the relative error of ONE component d_i -- an ulp of x_i over |d_i| = ulp * stpmx /
    |bound - x_i| -- so its tolerance grows with stpmx itself (1e-9 * max(1, stpmx))."""
    assert got.task_s == exp.task_s
    assert bytes(got.csave.tobytes()).rstrip() == bytes(exp.csave.tobytes()).rstrip()
    gi, ei = got.isave[21:44].copy(), exp.isave[21:44].copy()
    gi[2] = ei[2] = 0                     # isave(24): iteration-file unit
    assert np.array_equal(gi, ei), "counters differ: %s vs %s" % (gi, ei)
    assert np.array_equal(got.lsave != 0, exp.lsave != 0)
    gd, ed = got.dsave.copy(), exp.dsave.copy()
    gd[TIME_D] = ed[TIME_D] = 0
    for k in range(29):
        if not (np.isfinite(gd[k]) and np.isfinite(ed[k])):   # (see nrm_close)
            assert (np.isnan(gd[k]) and np.isnan(ed[k])) or gd[k] == ed[k], "dsave(%d): %r vs %r" % (k + 1, gd[k], ed[k])
            continue
        s = max(abs(ed[k]), 1e-300)
        # (dsave(28:29) = dcsrch's width, width1 = stpmax - stpmin and twice that: stpmx again)
        tol = 1e-9 * (max(1.0, abs(ed[11])) if (stpmx_cond and k in (11, 27, 28)) else 1.0)
        assert abs(gd[k] - ed[k]) <= tol * s + 1e-12 * max(1.0, abs(float(exp.f[0]))), \
            "dsave(%d): %r vs %r" % (k + 1, gd[k], ed[k])
    nrm_close(got.x, exp.x, rtol, "x")
    nrm_close(got.g, exp.g, rtol, "g")
    nrm_close(got.f, exp.f, rtol, "f")
    off = po.wa_offsets(n, m)
    col = int(exp.isave[27])

    def seg(s, name):
        o, ln = off[name]
        return s.wa[o:o + ln]
    for name in ("z", "r", "d", "t", "xp"):
        if name in skip:
            continue
        fl = float(np.max(np.abs(exp.x))) * 1e-3
        nrm_close(seg(got, name), seg(exp, name), rtol, name, floor=fl)
    nrm_close(seg(got, "ws"), seg(exp, "ws"), rtol, "ws")
    nrm_close(seg(got, "wy"), seg(exp, "wy"), rtol, "wy")
    if col > 0:
        for name in ("sy", "ss"):
            a = seg(got, name).reshape(m, m, order="F")[:col, :col]
            b = seg(exp, name).reshape(m, m, order="F")[:col, :col]
            a, b = (np.tril(a), np.tril(b)) if name == "sy" else (np.triu(a), np.triu(b))
            nrm_close(a, b, 1e-9, name)
        a = np.triu(seg(got, "wt").reshape(m, m, order="F")[:col, :col])
        b = np.triu(seg(exp, "wt").reshape(m, m, order="F")[:col, :col])
        nrm_close(a, b, 1e-7, "wt")
        # why 1e-7 and not 1e-10: wt is the Cholesky factor of T = theta S'S + L D^-1 L' (formt,
        # :1926-1966), formed from sy / ss, whose entries are n-term sums (reassociated here).
        # A factor moves by up to ~ cond(T) x the relative change of T: check the tolerance
        # against THAT -- the error of wt must be explained by cond(T) = cond(wt)^2 times the
        # error of its inputs (or by rounding itself, 1e-15), with a modest constant.
        sy_g, sy_e = (np.tril(seg(s_, "sy").reshape(m, m, order="F")[:col, :col]) for s_ in (got, exp))
        ss_g, ss_e = (np.triu(seg(s_, "ss").reshape(m, m, order="F")[:col, :col]) for s_ in (got, exp))
        # (theta = y'y / s'y scales S'S in T: its sums carry the same reassociation error)
        th_err = abs(float(got.dsave[0]) - float(exp.dsave[0])) / max(abs(float(exp.dsave[0])), 1e-300)
        e_in = max(rel_err(sy_g, sy_e), rel_err(ss_g, ss_e), th_err, 1e-15)
        cond_t = float(np.linalg.cond(b)) ** 2 if col > 1 else 1.0
        amp = rel_err(a, b) / (cond_t * e_in)
        COND_SEEN["wt"] = max(COND_SEEN["wt"], amp)
        assert amp <= 50.0, "wt: error %.2e, cond(T) %.2e, input error %.2e" % (rel_err(a, b), cond_t, e_in)
        # formk state: WN1 (kept incrementally, reference :1735-1851) and the factored WN
        ga = seg(got, "snd").reshape(2 * m, 2 * m, order="F")
        ea = seg(exp, "snd").reshape(2 * m, 2 * m, order="F")
        scale = float(np.max(np.abs(ea))) * 1e-6
        nrm_close(np.tril(ga[:col, :col]), np.tril(ea[:col, :col]), 1e-8, "wn1 Y'ZZ'Y", floor=scale)
        nrm_close(np.tril(ga[m:m + col, m:m + col]), np.tril(ea[m:m + col, m:m + col]), 1e-8,
                  "wn1 S'AA'S", floor=scale)
        nrm_close(ga[m:m + col, :col], ea[m:m + col, :col], 1e-8, "wn1 L_a+R_z", floor=scale)
        gw = seg(got, "wn").reshape(2 * m, 2 * m, order="F")[:2 * col, :2 * col]
        ew = seg(exp, "wn").reshape(2 * m, 2 * m, order="F")[:2 * col, :2 * col]
        nrm_close(np.triu(gw), np.triu(ew), 1e-6, "wn")
        # the same argument for the factored K of formk (:1856-1906): wn holds its triangular
        # factor; its error against cond(K) = cond(factor)^2 times the error of WN1 (and of theta,
        # sy through the assembly)
        e_in = max(rel_err(np.tril(ga[:col, :col]), np.tril(ea[:col, :col]), scale),
                   rel_err(np.tril(ga[m:m + col, m:m + col]), np.tril(ea[m:m + col, m:m + col]), scale),
                   rel_err(ga[m:m + col, :col], ea[m:m + col, :col], scale), e_in, 1e-15)
        cond_k = float(np.linalg.cond(np.triu(ew))) ** 2
        amp = rel_err(np.triu(gw), np.triu(ew)) / (cond_k * e_in)
        COND_SEEN["wn"] = max(COND_SEEN["wn"], amp)
        assert amp <= 50.0, "wn: error %.2e, cond(K) %.2e, input error %.2e" % (
            rel_err(np.triu(gw), np.triu(ew)), cond_k, e_in)
        # cauchy / cmprlb / subsm scratch wa(8m) (:617-619): at a first FG_LNSRCH return
        # wa(1:2m) = K^-1 W'Zr of subsm (through two triangular solves with the factored K:
        # the tolerance of wn), wa(2m+1:4m) = c = W'(xcp - x), wa(4m+1:6m) = the last wbp,
        # wa(6m+1:8m) = the last M*v product of the walk
        if exp.task_s.startswith("FG_LN") and int(exp.isave[35]) == 1 and "wa8m" not in skip:
            w8g, w8e = seg(got, "wa8m"), seg(exp, "wa8m")
            cs = float(np.max(np.abs(w8e[2 * m:4 * m])))
            nrm_close(w8g[2 * m:2 * m + 2 * col], w8e[2 * m:2 * m + 2 * col], 1e-9, "wa8m c", floor=cs * 1e-3 + 1e-300)
            # (wv comes out of two triangular solves with the factored K: beyond 1e-6 its error must be
            #  explained the way wn's is -- cond(K) times the error of what went in; n = 6 with 13 pairs
            #  stored makes K singular to working precision)
            wv_tol = max(1e-6, 50.0 * cond_k * e_in)
            kinv = float(np.linalg.norm(np.linalg.inv(np.triu(ew)), 2)) ** 2
            nseg_ = float(exp.isave[32])
            gmax = float(np.max(np.abs(exp.g)))
            wmax = max(float(np.max(np.abs(seg(exp, "ws")))), float(np.max(np.abs(seg(exp, "wy")))))
            # (2 col > n: more stored columns than variables -- K = the 2col x 2col middle matrix of a rank <= n
            #  update is singular by construction and kept apart from that by rounding alone; W'Z r, which the
            #  library takes in closed form from the walk's p and the reference from a loop over the rows, then
            #  cancels to nothing in two different ways.  Fuzz seed 911214: n = 5, col = 4 at convergence,
            #  |K^-1| = 6e18, wv of size 16 equal to 1.1e-6, d = z - x -- what wv is FOR -- equal in every digit.
            #  There the check of wv is the check of d, z above.)
            if 2 * col <= n:
                nrm_close(w8g[:2 * col], w8e[:2 * col], wv_tol, "wa8m wv",
                          floor=max(1e-9 * max(1.0, float(np.max(np.abs(exp.x))), gmax),
                                    (4.0 + nseg_) * np.finfo(np.float64).eps * n * gmax * wmax * max(1.0, kinv) / wv_tol))
            # (floor: at convergence, and for n of a few variables, wv is rounding noise of sums of |g|-sized terms
            #  -- W'Z r: n products |r_i| |w_ij| <= gmax wmax each, summed in another order: 4 eps n gmax wmax,
            #  carried through the two triangular solves: |K^-1| <= |factor^-1|^2; c = W'(xcp - x) behind it is
            #  accumulated over the walk's segments from a p that starts at n gmax wmax and cancels: + nseg)
            if "wbp" not in skip:
                # (wbp = row of W of the LAST variable the walk fixed, v = M wbp: work vectors nothing reads
                #  afterwards.  A production context crosses a WHOLE group of equal breakpoints in index order,
                #  the reference in heap order -- same group, another last member)
                nrm_close(w8g[4 * m:4 * m + 2 * col], w8e[4 * m:4 * m + 2 * col], rtol, "wa8m wbp")
                nrm_close(w8g[6 * m:6 * m + 2 * col], w8e[6 * m:6 * m + 2 * col], 1e-7, "wa8m v")
    iw_g = got.iwa[n:2 * n]
    iw_e = exp.iwa[n:2 * n]
    if check_iwhere:
        assert np.array_equal(iw_g, iw_e), "iwhere differs at %s" % np.nonzero(iw_g != iw_e)[0][:8]
    if check_lists:
        assert np.array_equal(got.iwa[:n], exp.iwa[:n]), "Index differs"
        nenter, ileave = int(exp.isave[40]), int(exp.isave[39])
        if ileave >= 1 and check_indx2:   # freev has run
            assert np.array_equal(got.iwa[2 * n:2 * n + nenter], exp.iwa[2 * n:2 * n + nenter])
            assert np.array_equal(got.iwa[2 * n + ileave - 1:], exp.iwa[2 * n + ileave - 1:])


def oracle_snapshots(po, p, max_calls, on_new_x=None):
    snaps = []
    po.run(po.Engine("oracle"), p, max_calls=max_calls, snapshot=lambda k, s: snaps.append(s.copy()),
           on_new_x=on_new_x)
    return snaps


def gpu_one_call(env, p, s_in):
    """Load caller state s_in into a fresh device context, make ONE setulb call through the
    C ABI (device-pointer form) and return the resulting caller state."""
    po, torch, la = env["po"], env["torch"], env["la"]
    s = s_in.copy()
    t = s.task_s
    if t.startswith("FG"):
        s.f[0] = p.fg(s.x, s.g)
    sol = la.DeviceSolver(p.n, p.m, mirror_index=True)
    try:
        x, g = _dev(torch, s.x), _dev(torch, s.g)
        l, u, nbd = _dev(torch, p.l), _dev(torch, p.u), _dev(torch, p.nbd.astype(np.int32))
        if not t.startswith("START"):
            sol.import_state(s.wa, s.iwa, s.isave)
        sol.task[:] = s.task
        sol.csave[:] = s.csave
        sol.lsave[:] = s.lsave
        sol.isave[:] = s.isave
        sol.dsave[:] = s.dsave
        sol.f[0] = s.f[0]
        sol.setulb(x, l, u, nbd, g, p.factr, p.pgtol)
        torch.cuda.synchronize()
        wa, iwa = sol.export_state()
        out = po.State(p.n, p.m, x.cpu().numpy(), g.cpu().numpy(), sol.f.copy(), wa, iwa,
                       sol.task.copy(), sol.csave.copy(), sol.lsave.copy(), sol.isave.copy(),
                       sol.dsave.copy())
    finally:
        sol.close()
    return s, out


ONE_STEP_CASES = [
    ("rosenbrock25", dict(kind="ros", n=25, m=5, factr=1e7, pgtol=1e-5), 52, 1),
    ("rosenbrock1000", dict(kind="ros", n=1000, m=10, factr=0.0, pgtol=0.0), 60, 1),
    ("quad1000", dict(kind="quad", n=1000, m=10), 70, 1),
    ("quadmix4099", dict(kind="quadmix", n=4099, m=10), 60, 1),
    ("quadmix777_m3", dict(kind="quadmix", n=777, m=3), 40, 1),
    ("quad20000_m17", dict(kind="quad", n=20000, m=17), 44, 2),
    ("quadmix3001_m25", dict(kind="quadmix", n=3001, m=25), 70, 3),   # col > 20: from-scratch formk path
]


def make_problem(po, spec):
    if spec["kind"] == "ros":
        return po.problem_rosenbrock(spec["n"], spec["m"], spec["factr"], spec["pgtol"])
    return po.problem_quadratic(spec["n"], spec["m"], mixed_nbd=spec["kind"] == "quadmix")


@pytest.mark.parametrize("name,spec,ncalls,stride", ONE_STEP_CASES, ids=[c[0] for c in ONE_STEP_CASES])
def test_one_step_parity_from_identical_state(env, name, spec, ncalls, stride):
    """Every setulb return of the oracle trajectory is used as the INPUT state of one GPU call;
    the GPU output must equal the oracle's next state: integers exactly, floats to 1e-10."""
    po = env["po"]
    p = make_problem(po, spec)
    snaps = oracle_snapshots(po, p, ncalls)
    # call 0 is START itself
    fresh = po.State.fresh(p)
    _, out0 = gpu_one_call(env, p, fresh)
    compare_states(out0, snaps[0], p.n, p.m, po, check_lists=False)
    tested = 0
    for k in range(0, len(snaps) - 1, stride):
        t = snaps[k].task_s
        if not (t.startswith("FG") or t.startswith("NEW_X")):
            continue
        _, out = gpu_one_call(env, p, snaps[k])
        compare_states(out, snaps[k + 1], p.n, p.m, po)
        tested += 1
    assert tested >= min(10, (len(snaps) - 1) // stride)


PRODUCTION_CASES = [
    ("quad1000", dict(kind="quad", n=1000, m=10), 70),
    ("quadmix4099", dict(kind="quadmix", n=4099, m=10), 64),
    ("rosenbrock1000", dict(kind="ros", n=1000, m=10, factr=0.0, pgtol=0.0), 60),
    ("quadmix777_m3", dict(kind="quadmix", n=777, m=3), 44),
    ("quad20011_m7", dict(kind="quad", n=20011, m=7), 40),
]


@pytest.mark.parametrize("name,spec,ncalls", PRODUCTION_CASES, ids=[c[0] for c in PRODUCTION_CASES])
def test_production_path_two_step_parity(env, name, spec, ncalls):
    """The path bench.py times -- a DEFAULT context (no mirroring of the reference's lists): the
    update pass runs speculatively as the evaluation of the first trial point, the accepted pair
    stays pending, the Cauchy point is kept in functional form -- checked per array, not only
    through its counters.  Every first-trial FG_LNSRCH return of the oracle's trajectory is
    imported into a default context; call 1 (the trial is accepted -> NEW_X) and call 2 (the whole
    next iteration up to its first trial point) must reproduce the oracle's states after the same
    two calls: x, g, f to 1e-10, iwhere / Index / all counters exactly, z, d, t, r, Ws, Wy, the
    m x m matrices, WN1 / WN, wa(8m).  Differences by design, documented in DESIGN.md section 7: at the
    NEW_X return iwhere already holds the pattern of the next cauchy scan (the reference updates
    it one call later, :1284-1291), and xp / the enter-leave half of Indx2 are not materialised
    (dead outside the call that makes them); export_state itself is read-only."""
    po, torch, la = env["po"], env["torch"], env["la"]
    p = make_problem(po, spec)
    snaps = oracle_snapshots(po, p, ncalls)
    tested = 0
    for k in range(len(snaps) - 2):
        s0, s1, s2 = snaps[k], snaps[k + 1], snaps[k + 2]
        if not (s0.task_s.startswith("FG_LN") and int(s0.isave[35]) == 1 and s1.task_s.startswith("NEW_X")
                and s2.task_s.startswith("FG_LN")):
            continue
        s = s0.copy()
        s.f[0] = p.fg(s.x, s.g)
        sol = la.DeviceSolver(p.n, p.m)            # default flags: the production path
        try:
            x, g = _dev(torch, s.x), _dev(torch, s.g)
            l, u, nbd = _dev(torch, p.l), _dev(torch, p.u), _dev(torch, p.nbd.astype(np.int32))
            sol.import_state(s.wa, s.iwa, s.isave)
            sol.task[:] = s.task
            sol.csave[:] = s.csave
            sol.lsave[:] = s.lsave
            sol.isave[:] = s.isave
            sol.dsave[:] = s.dsave
            sol.f[0] = s.f[0]
            st0 = sol.stats()

            def snapshot():
                torch.cuda.synchronize()
                wa, iwa = sol.export_state()
                return po.State(p.n, p.m, x.cpu().numpy(), g.cpu().numpy(), sol.f.copy(), wa, iwa,
                                sol.task.copy(), sol.csave.copy(), sol.lsave.copy(), sol.isave.copy(),
                                sol.dsave.copy())
            sol.setulb(x, l, u, nbd, g, p.factr, p.pgtol)
            out1 = snapshot()
            # iwhere: compared against the oracle's state after the NEXT call's scan below
            compare_states(out1, s1, p.n, p.m, po, skip=("xp",), check_indx2=False, check_iwhere=False)
            launches_before = sol.stats()["launches"]
            sol.setulb(x, l, u, nbd, g, p.factr, p.pgtol)
            out2 = snapshot()
            compare_states(out2, s2, p.n, p.m, po, skip=("xp",), check_indx2=False)
            assert sol.stats()["launches"] > launches_before > st0["launches"]
        finally:
            sol.close()
        tested += 1
    assert tested >= 8, tested


def test_state_round_trip_default_context(env):
    """export_state / import_state on a DEFAULT context across NEW_X boundaries with bound changes
    (ADVICE r1): a run that is exported at iteration k, imported into a fresh context and
    continued must produce the same trajectory as the uninterrupted run -- the free-set
    membership travels through Index, rebuilt from the device's membership bytes."""
    po, torch, la = env["po"], env["torch"], env["la"]
    p = po.problem_quadratic(4099, 6, mixed_nbd=True)

    def drive(sol, x, g, l, u, nbd, until_iter, rows):
        while True:
            t = sol.setulb(x, l, u, nbd, g, 0.0, 0.0)
            if t.startswith("FG"):
                sol.f[0] = sol.objective(0, x, g)
            elif t.startswith("NEW_X"):
                rows.append((int(sol.isave[29]), int(sol.isave[33]), int(sol.isave[32]), int(sol.isave[37]),
                             int(sol.isave[39]), int(sol.isave[40]), float(sol.f[0])))
                if sol.isave[29] >= until_iter:
                    return
            else:
                return

    def tensors():
        return (torch.from_numpy(p.x0.copy()).cuda(), torch.zeros(p.n, dtype=torch.float64, device="cuda"),
                torch.from_numpy(p.l).cuda(), torch.from_numpy(p.u).cuda(),
                torch.from_numpy(p.nbd.astype(np.int32)).cuda())
    ref_rows = []
    sol = la.DeviceSolver(p.n, p.m)
    x, g, l, u, nbd = tensors()
    drive(sol, x, g, l, u, nbd, 14, ref_rows)
    sol.close()
    for cut in (3, 7, 9):
        rows = []
        a = la.DeviceSolver(p.n, p.m)
        x, g, l, u, nbd = tensors()
        drive(a, x, g, l, u, nbd, cut, rows)
        torch.cuda.synchronize()
        wa, iwa = a.export_state()
        assert int(np.count_nonzero(iwa[:p.n])) == p.n                 # Index is a permutation of 1..n
        assert sorted(iwa[:p.n].tolist()) == list(range(1, p.n + 1))
        b = la.DeviceSolver(p.n, p.m)
        b.import_state(wa, iwa, a.isave)
        for name in ("task", "csave", "lsave", "isave", "dsave", "f"):
            getattr(b, name)[:] = getattr(a, name)
        a.close()
        drive(b, x, g, l, u, nbd, 14, rows)
        b.close()
        assert [r[:6] for r in rows] == [r[:6] for r in ref_rows], (cut, rows, ref_rows)
        for r, q in zip(rows, ref_rows):
            assert r[6] == pytest.approx(q[6], rel=1e-12)
    # garbage in Index must be refused, not indexed with
    bad = la.DeviceSolver(p.n, p.m)
    iwa2 = iwa.copy()
    iwa2[0] = p.n + 5
    with pytest.raises(la.LbfgsbError):
        bad.import_state(wa, iwa2, a.isave)
    bad.close()


def run_host_api(env, p, max_calls, on_new_x=None, iprint=-1, iteration_file=None, mirror=True):
    """The drop-in form: reference argument list, host arrays (lbfgsb_amd.setulb)."""
    po, la = env["po"], env["la"]
    s = po.State.fresh(p)
    nbd = p.nbd.astype(np.int32)
    snaps = []
    for _ in range(max_calls):
        la.setulb(p.n, p.m, s.x, p.l, p.u, nbd, s.f, s.g, p.factr, p.pgtol, s.wa, s.iwa, s.task,
                  iprint, s.csave, s.lsave, s.isave, s.dsave, iteration_file=iteration_file,
                  mirror=mirror)
        snaps.append(s.copy())
        t = s.task_s
        if t.startswith("FG"):
            s.f[0] = p.fg(s.x, s.g)
        elif t.startswith("NEW_X"):
            if on_new_x is not None:
                stop = on_new_x(s)
                if stop:
                    s.task[:] = po.pad60(stop)
        else:
            break
    return snaps


def driver_stop(lim):
    def rule(s):
        if s.isave[33] >= lim:
            return "STOP: TOTAL NO. of f AND g EVALUATIONS EXCEEDS LIMIT"
        if s.dsave[12] <= 1.0e-10 * (1.0 + abs(float(s.f[0]))):
            return "STOP: THE PROJECTED GRADIENT IS SUFFICIENTLY SMALL"
        return None
    return rule


def test_driver1_trajectory_drop_in(env, tmp_path):
    """reference test/driver1.f90 through the host-pointer ABI: the integer columns of
    test/OUTPUTS/iterate.dat (it nf nseg nact itls) exactly, 23 iterations / 28 evaluations /
    47 segments, floats to 6 digits, and the iteration file written by the library."""
    po = env["po"]
    p = po.problem_rosenbrock(25, 5, 1e7, 1e-5)
    itf = str(tmp_path / "driver1_output.txt")
    snaps = run_host_api(env, p, 200, iprint=1, iteration_file=itf)
    last = snaps[-1]
    assert last.task_s == "CONVERGENCE: REL_REDUCTION_OF_F_<=_FACTR*EPSMCH"
    assert (last.isave[29], last.isave[33], last.isave[21], last.isave[25]) == (23, 28, 47, 0)
    assert float(last.f[0]) == pytest.approx(1.0834900834300614e-09, rel=1e-6)
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_outputs",
                        "iterate.dat")

    def rows(path):
        out = []
        with open(path) as fh:
            for line in fh:
                tk = line.split()
                if len(tk) == 10 and tk[0].isdigit() and tk[2].isdigit():
                    out.append(tk)
        return out
    a, b = rows(itf), rows(gold)
    assert len(a) == len(b) == 23
    for ra, rb in zip(a, b):
        assert ra[:6] == rb[:6], (ra, rb)          # it nf nseg nact sub itls
        for ca, cb in zip(ra[6:], rb[6:]):
            assert float(ca.replace("D", "E")) == pytest.approx(float(cb.replace("D", "E")), rel=2e-3)


@pytest.mark.parametrize("case,lim", [("driver2", 99), ("driver3", 900)])
def test_driver23_trajectory_drop_in(env, case, lim):
    """driver2 (n=25) / driver3 (n=1000) settings with their user stops: per-iteration integer
    state equal to the reference's golden trajectory while f is far above its noise floor."""
    po = env["po"]
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", case + "_traj.npz"))
    p = po.problem_rosenbrock(int(z["n"]), int(z["m"]), 0.0, 0.0)
    snaps = run_host_api(env, p, 400, on_new_x=driver_stop(lim), mirror=False)
    gold_new_x = [k for k in range(z["f"].shape[0]) if bytes(z["task"][k].tobytes()).startswith(b"NEW_X")]
    mine_new_x = [k for k, s in enumerate(snaps) if s.task_s.startswith("NEW_X")]
    checked = 0
    for ka, kb in zip(mine_new_x, gold_new_x):
        fa, fb = float(snaps[ka].f[0]), float(z["f"][kb])
        if fb < 1e-9:
            break
        ia, ib = snaps[ka].isave, z["isave"][kb]
        for slot in (29, 33, 32, 38, 35):           # iter nfgv nseg nact ifun
            assert ia[slot] == ib[slot], (slot, ia[slot], ib[slot])
        assert fa == pytest.approx(fb, rel=1e-6)
        checked += 1
    assert checked >= 20
    assert snaps[-1].task_s.startswith("STOP: THE PROJECTED GRADIENT IS SUFFICIENTLY SMALL")


def test_kernel_projgr_bit_exact(env):
    po, torch, la = env["po"], env["torch"], env["la"]
    rng = np.random.default_rng(7)
    for n in (1, 2, 3, 255, 4097, 100003):
        x = rng.standard_normal(n)
        g = rng.standard_normal(n) * 3
        l = x - rng.random(n) * (rng.random(n) > 0.3)
        u = x + rng.random(n) * (rng.random(n) > 0.3)
        nbd = rng.integers(0, 4, n).astype(np.int32)
        eo = po.Engine("oracle")
        eo.lib.lbo_projgr.restype = __import__("ctypes").c_double
        import ctypes as C
        eo.lib.lbo_projgr.argtypes = [C.c_int] + [C.c_void_p] * 5
        want = eo.lib.lbo_projgr(n, l.ctypes.data, u.ctypes.data, nbd.ctypes.data, x.ctypes.data, g.ctypes.data)
        sol = la.DeviceSolver(n, 3)
        got = sol.projgr(_dev(torch, x), _dev(torch, l), _dev(torch, u), _dev(torch, nbd), _dev(torch, g))
        sol.close()
        assert got == want            # a max-reduction is exact in any order


def test_kernel_wtv_against_float128(env):
    """The WS/WY matvec for every column count 1..m, circular head, ragged n."""
    po, torch, la = env["po"], env["torch"], env["la"]
    rng = np.random.default_rng(11)
    for n, m in ((5, 3), (1000, 5), (4099, 10), (30001, 20), (2049, 32)):
        ws = rng.standard_normal((m, n))
        wy = rng.standard_normal((m, n))
        v = rng.standard_normal(n)
        sol = la.DeviceSolver(n, m)
        sol.set_w(ws, wy)
        vd = _dev(torch, v)
        for col in sorted({1, 2, m // 2 + 1, m}):
            for head in (1, m):
                got = sol.wtv(vd, col, head)
                cols = [(head - 1 + j) % m for j in range(col)]
                want = np.concatenate([wy[cols].astype(np.longdouble) @ v.astype(np.longdouble),
                                       ws[cols].astype(np.longdouble) @ v.astype(np.longdouble)])
                bound = np.concatenate([np.abs(wy[cols]) @ np.abs(v), np.abs(ws[cols]) @ np.abs(v)])
                assert np.all(np.abs(got - want.astype(np.float64)) <= 1e-13 * bound + 1e-300)
        # linearity (size-independent property): W'(a v1 + v2) = a W'v1 + W'v2
        v2 = rng.standard_normal(n)
        a = 0.37
        lhs = sol.wtv(_dev(torch, a * v + v2), m, 1)
        rhs = a * sol.wtv(vd, m, 1) + sol.wtv(_dev(torch, v2), m, 1)
        assert np.allclose(lhs, rhs, rtol=1e-11, atol=1e-11 * np.sqrt(n))
        sol.close()


@pytest.mark.parametrize("real32", [False, True], ids=["fp64", "fp32"])
def test_kernel_formk_gram_against_numpy(env, real32):
    """formk's inner products from scratch (src/lbfgsb.f90:1756-1851: Y'ZZ'Y over the free rows,
    S'AA'S over the active rows, L_a + R_z mixed) for every column count / circular head / ragged
    n, both the quad kernel (col <= 10) and the LDS-tile kernel (col > 10), against numpy in
    extended precision."""
    po, torch, la = env["po"], env["torch"], env["la"]
    rng = np.random.default_rng(23)
    real = np.float32 if real32 else np.float64
    for n, m in ((1, 3), (7, 5), (1000, 5), (4099, 10), (30001, 10), (2049, 20)):
        ws = rng.standard_normal((m, n)).astype(real)
        wy = rng.standard_normal((m, n)).astype(real)
        iw = rng.integers(-3, 4, n).astype(np.int32)
        sol = la.DeviceSolver(n, m, real32=real32)
        sol.set_w(ws, wy)
        sol.set_iwhere(iw)
        free = (iw <= 0)
        for col in sorted({1, 2, m // 2 + 1, m}):
            for head in (1, m):
                got = sol.formk_gram(col, head)
                cols = [(head - 1 + j) % m for j in range(col)]
                Y = wy[cols].astype(np.longdouble)
                S = ws[cols].astype(np.longdouble)
                Yf, Sa, Sf = Y * free, S * (~free), S * free
                tri = col * (col + 1) // 2
                want = np.zeros(2 * col * col + col, np.longdouble)
                bound = np.zeros_like(want)
                for i in range(col):
                    for j in range(i + 1):
                        want[i * (i + 1) // 2 + j] = Yf[i] @ Y[j]
                        bound[i * (i + 1) // 2 + j] = np.abs(Yf[i]) @ np.abs(Y[j])
                        want[tri + i * (i + 1) // 2 + j] = Sa[i] @ S[j]
                        bound[tri + i * (i + 1) // 2 + j] = np.abs(Sa[i]) @ np.abs(S[j])
                    for j in range(col):
                        a = Sa[i] if i > j else Sf[i]
                        want[2 * tri + i * col + j] = a @ Y[j]
                        bound[2 * tri + i * col + j] = np.abs(a) @ np.abs(Y[j])
                assert np.all(np.abs(got - want.astype(np.float64)) <= 1e-13 * bound.astype(np.float64) + 1e-300), \
                    (n, m, col, head)
        sol.close()


def test_edge_cases(env):
    """n=1, unconstrained (cauchy skipped after the first update), all variables fixed,
    zero gradient at start, invalid nbd and l>u."""
    po = env["po"]
    # unconstrained quadratic: nbd = 0 everywhere
    p = po.problem_quadratic(513, 4)
    p.nbd[:] = 0
    a = oracle_snapshots(po, p, 40)
    b = run_host_api(env, p, 40)
    assert len(a) == len(b)
    for sa, sb in zip(a, b):
        assert sa.task_s == sb.task_s
        assert np.array_equal(sa.isave[21:44][[6, 8, 12, 11, 16]], sb.isave[21:44][[6, 8, 12, 11, 16]])
        nrm_close(sb.x, sa.x, 1e-7, "x (trajectory)")
    # n = 1
    p1 = po.problem_quadratic(1, 3)
    a = oracle_snapshots(po, p1, 30)
    b = run_host_api(env, p1, 30)
    assert [s.task_s for s in a] == [s.task_s for s in b]
    nrm_close(b[-1].x, a[-1].x, 1e-9, "x n=1")
    # every variable fixed: l = u = x0
    pf = po.problem_quadratic(300, 5)
    pf.l[:] = 0.25
    pf.u[:] = 0.25
    a = oracle_snapshots(po, pf, 10)
    b = run_host_api(env, pf, 10)
    assert [s.task_s for s in a] == [s.task_s for s in b]
    assert np.array_equal(b[-1].x, a[-1].x)
    # errors detected by errclb (reference :1601-1643)
    pe = po.problem_quadratic(100, 5)
    pe.nbd[17] = 7
    b = run_host_api(env, pe, 3)
    assert b[-1].task_s == "ERROR: INVALID NBD" and b[-1].isave[21 + 13] == 0
    pe = po.problem_quadratic(100, 5)
    pe.l[40] = 2.0
    b = run_host_api(env, pe, 3)
    assert b[-1].task_s == "ERROR: NO FEASIBLE SOLUTION"


def test_device_objectives_match_oracle(env):
    po, torch, la = env["po"], env["torch"], env["la"]
    rng = np.random.default_rng(3)
    for n in (7, 1001, 65537):
        x = rng.standard_normal(n)
        g = np.zeros(n)
        sol = la.DeviceSolver(n, 3)
        xd, gd = _dev(torch, x), _dev(torch, g)
        for kind, fn in ((0, po.quadratic_fg), (1, po.rosenbrock_fg)):
            f_dev = sol.objective(kind, xd, gd)
            f_cpu = fn(x, g)
            assert f_dev == pytest.approx(f_cpu, rel=1e-12)
            assert np.array_equal(gd.cpu().numpy(), g)      # elementwise: bit-exact
        sol.close()


def test_full_size_quadratic_n1e6_against_oracle(env):
    """BASELINE.json configs[1]: separable bounded quadratic, n = 1e6, m = 10, on-device
    objective, 30 iterations.  Iteration 1 walks ~976,721 Cauchy segments (full breakpoint
    sort); integer state must equal the oracle's at every iteration, f to 1e-9.  Anchors pinned by the reference itself
    (BASELINE.md section 2): nseg(it1) = 976721, nfree(it1) = 23280, f(it1) = 8.2541454907951783E+06."""
    po, torch, la = env["po"], env["torch"], env["la"]
    n, m, iters = 1_000_000, 10, 30
    p = po.problem_quadratic(n, m)
    rows_o = []
    po.run(po.Engine("oracle"), p, max_iter=iters,
           snapshot=lambda k, s: rows_o.append((int(s.isave[29]), int(s.isave[33]), int(s.isave[32]),
                                                int(s.isave[37]), float(s.f[0]))) if s.task_s.startswith("NEW_X") else None)
    sol = la.DeviceSolver(n, m)
    x = torch.zeros(n, dtype=torch.float64, device="cuda")
    g = torch.zeros_like(x)
    l, u = torch.full_like(x, -1.0), torch.full_like(x, 1.0)
    nbd = torch.full((n,), 2, dtype=torch.int32, device="cuda")
    rows_g = []
    while True:
        t = sol.setulb(x, l, u, nbd, g, 0.0, 0.0)
        if t.startswith("FG"):
            sol.f[0] = sol.objective(0, x, g)
        elif t.startswith("NEW_X"):
            rows_g.append((int(sol.isave[29]), int(sol.isave[33]), int(sol.isave[32]),
                           int(sol.isave[37]), float(sol.f[0])))
            if sol.isave[29] >= iters:
                break
        else:
            break
    st = sol.stats()
    closed_steps, three_steps, _ = sol.path_counts()
    sol.close()
    assert rows_o[0][2] == 976721 and rows_o[0][3] == 23280
    assert rows_o[0][4] == pytest.approx(8.2541454907951783e06, rel=1e-13)
    # 30 iterations: from iteration 11 on the memory is full (col = m) and the iteration is the
    # bench's own steady state -- two passes over W, W'Z r in closed form, lean stores, pending
    # pair.  More anchors printed by the reference itself: it 10, 20, 30
    assert rows_o[9][2:4] == (39, 499959) and rows_o[29][1] == 32
    for k, fref in ((1, 4.4408007558922265e06), (2, 4.2654335281014517e06), (9, 4.2089636688019084e06),
                    (19, 4.2086430506394058e06), (29, 4.2086404848337891e06)):
        assert rows_o[k][4] == pytest.approx(fref, rel=1e-13)
        assert rows_g[k][4] == pytest.approx(fref, rel=1e-9)
    assert len(rows_g) == len(rows_o) == iters
    for a, b in zip(rows_g, rows_o):
        assert a[:4] == b[:4], (a, b)
        assert a[4] == pytest.approx(b[4], rel=1e-9)
    assert st["cauchy_fullsorts"] >= 1
    assert closed_steps >= 15, (closed_steps, three_steps)   # the closed form really was the path taken


@pytest.mark.parametrize("kind,n,m,upto", [("quad", 1_000_000, 10, 34), ("quadmix", 1_000_003, 7, 30),
                                           ("ros", 1_000_000, 10, 30), ("quad", 5_000_000, 10, 30)])
def test_one_step_parity_at_size(env, kind, n, m, upto):
    """The 1e-10 bar of north_star ("over identical iterates") at SIZE (it used to be
    enforced by the one-step tests at n <= 20 000 only, the full-size tests being trajectory tests with
    drift allowances).  The oracle runs the problem at n = 1e6 / 5e6; several of its setulb returns -- the
    first walk (0.977 n segments), iterations while the memory fills, the steady state with col = m --
    are imported into a PRODUCTION context (default flags: speculative update pass, pending pair,
    closed-form W'Z r, functional Cauchy point) and stepped once: task, every counter, iwhere exactly;
    x, g, f, d, t, r, Ws, Wy, the m x m matrices at 1e-10 / the tolerances of compare_states -- sums over
    10^6 - 5 10^6 rows against the oracle's sequential ones."""
    po, torch, la = env["po"], env["torch"], env["la"]
    if kind == "ros":
        p = po.problem_rosenbrock(n, m, 0.0, 0.0)
    else:
        p = po.problem_quadratic(n, m, mixed_nbd=(kind == "quadmix"))
    want = sorted(set([0, 1, 2, 3, 6, 7, upto - 12, upto - 11, upto - 4, upto - 3, upto - 2]))
    keep = {}
    po.run(po.Engine("oracle"), p, max_calls=upto,
           snapshot=lambda k, s: keep.__setitem__(k, s.copy()) if (k in want or k - 1 in want) else None)
    l, u = _dev(torch, p.l), _dev(torch, p.u)
    nbd = _dev(torch, p.nbd.astype(np.int32))
    tested = 0
    for k in want:
        if k not in keep or k + 1 not in keep:
            continue
        s0, s1 = keep[k], keep[k + 1]
        t0 = s0.task_s
        if not (t0.startswith("FG") or t0.startswith("NEW_X")):
            continue
        s = s0.copy()
        if t0.startswith("FG"):
            s.f[0] = p.fg(s.x, s.g)
        sol = la.DeviceSolver(p.n, p.m)
        try:
            x, g = _dev(torch, s.x), _dev(torch, s.g)
            sol.import_state(s.wa, s.iwa, s.isave)
            for nm in ("task", "csave", "lsave", "isave", "dsave"):
                getattr(sol, nm)[:] = getattr(s, nm)
            sol.f[0] = s.f[0]
            sol.setulb(x, l, u, nbd, g, p.factr, p.pgtol)
            torch.cuda.synchronize()
            wa, iwa = sol.export_state()
            out = po.State(p.n, p.m, x.cpu().numpy(), g.cpu().numpy(), sol.f.copy(), wa, iwa, sol.task.copy(),
                           sol.csave.copy(), sol.lsave.copy(), sol.isave.copy(), sol.dsave.copy())
        finally:
            sol.close()
        # (production context: at a NEW_X return iwhere already holds the next scan's pattern -- and after a
        #  REJECTED first trial the pattern of the rejected point, until the next cauchy recomputes it; xp and
        #  the enter / leave half of Indx2 are not materialised -- DESIGN.md section 7)
        compare_states(out, s1, p.n, p.m, po, skip=("xp",), check_lists=False,
                       check_iwhere=out.task_s.startswith("FG_LN") and not t0.startswith("FG_LN"))
        tested += 1
    assert tested >= 6, tested


WIDE_CASES = [
    ("quadmix2003_m40", dict(kind="quadmix", n=2003, m=40), 110, 3),
    ("rosenbrock500_m33", dict(kind="ros", n=500, m=33, factr=0.0, pgtol=0.0), 90, 3),
    ("quad3001_m70", dict(kind="quad", n=3001, m=70), 100, 4),
]


@pytest.mark.parametrize("name,spec,ncalls,stride", WIDE_CASES, ids=[c[0] for c in WIDE_CASES])
def test_wide_memory_one_step_parity(env, name, spec, ncalls, stride):
    """m > 32 (the reference puts no limit on m, src/lbfgsb.f90:93-97): beyond the width of the fused
    kernels a context composes the iteration from unfused tile primitives (k_wide.hip, solver_wide.inl).
    Same bar as every one-step test: each return of the oracle's trajectory -- through col = 33 ... m, the
    shift of a full memory, rejected trials -- is the input of ONE GPU call (mirror context); output equal
    to the oracle's next state: integers and lists exactly, floats at the one-step tolerances."""
    po = env["po"]
    p = make_problem(po, spec)
    snaps = oracle_snapshots(po, p, ncalls)
    assert max(int(s.isave[27]) for s in snaps) > 32          # the memory really grows beyond the fused width
    tested = 0
    for k in range(0, len(snaps) - 1, stride):
        t = snaps[k].task_s
        if not (t.startswith("FG") or t.startswith("NEW_X")):
            continue
        _, out = gpu_one_call(env, p, snaps[k])
        compare_states(out, snaps[k + 1], p.n, p.m, po)
        tested += 1
    assert tested >= 20, tested


@pytest.mark.parametrize("m", [40, 70], ids=["m40", "m70"])
@pytest.mark.parametrize("real32", [False, True], ids=["fp64", "real32"])
@pytest.mark.parametrize("pp", [False, True], ids=["classic", "pingpong"])
def test_wide_tail_folded_into_the_r_pass_changes_no_row(env, pp, real32, m):
    """m > 32: cmprlb's start (r0 = -theta (xcp - x) - g) and subsm's tail (projected step, d = z - x, the line-search
    set-up, the first trial point) run inside the first / last tile of the r pass (tile_axpy_fused_kernel, option
    wide_tail) instead of as six vector kernels.  Per row that is the same arithmetic with the same roundings to the
    storage kind in between -- so the iterates must be the unfused route's BIT FOR BIT, in fp64 and in REAL32 (where
    every intermediate vector the unfused kernels store is rounded to fp32), through both entries, while the memory
    grows from 1 to m pairs and beyond (one tile, two, three; the shift of a full memory).  The same for the r pass
    as ONE launch over all columns with the pending pair committed by it (wide_r_pass_kernel, option wide_one, the
    default) against one launch per tile behind pair_commit_kernel: three routes, one set of bits."""
    po, torch, la = env["po"], env["torch"], env["la"]
    n, iters = 7001, m + 15
    rdt = torch.float32 if real32 else torch.float64
    p = po.problem_quadratic(n, m, mixed_nbd=True)

    def run(tail, one=1):
        sol = la.DeviceSolver(n, m, real32=real32, options={"wide_tail": tail, "wide_one": one})
        xs = [torch.from_numpy(p.x0.copy()).to(rdt).cuda(), torch.zeros(n, dtype=rdt, device="cuda")]
        gs = [torch.zeros_like(xs[0]), torch.zeros_like(xs[0])]
        l, u = torch.from_numpy(p.l).to(rdt).cuda(), torch.from_numpy(p.u).to(rdt).cuda()
        nbd = torch.from_numpy(p.nbd.astype(np.int32)).cuda()
        rows, cur = [], 0
        for _ in range(100000):
            if pp:
                t, cur = sol.setulb_pp(xs, l, u, nbd, gs, 0.0, 0.0)
            else:
                t = sol.setulb(xs[0], l, u, nbd, gs[0], 0.0, 0.0)
            if t.startswith("FG"):
                sol.f[0] = sol.objective(0, xs[cur], gs[cur])
                rows.append(("FG", int(sol.isave[35]), xs[cur].cpu().numpy().tobytes()))   # (ifun, the trial point)
            elif t.startswith("NEW_X"):
                rows.append(("NEW_X", int(sol.isave[29]), int(sol.isave[33]), int(sol.isave[32]), int(sol.isave[37]),
                             int(sol.isave[27]), float(sol.f[0]), xs[cur].cpu().numpy().tobytes()))
                if sol.isave[29] >= iters:
                    break
            else:
                break
        sol.close()
        return rows
    a, b, c = run(1), run(0), run(1, 0)
    assert len(a) == len(b) == len(c) and len(a) > iters
    assert max(r[5] for r in a if r[0] == "NEW_X") == m
    for k, (ra, rb, rc) in enumerate(zip(a, b, c)):
        assert ra == rb, (k, ra[:6], rb[:6])
        assert ra == rc, (k, ra[:6], rc[:6])


@pytest.mark.parametrize("pp", [False, True], ids=["classic", "pingpong"])
def test_wide_memory_trajectory_to_convergence(env, pp):
    """m = 48 on the production path (default context, both device-pointer entries), bounded quadratic with all
    four bound types, run to convergence with factr = pgtol = 0: integer columns equal the oracle's for at
    least 60 iterations, f to 1e-10 throughout, same final message."""
    po, torch, la = env["po"], env["torch"], env["la"]
    n, m = 5003, 48
    p = po.problem_quadratic(n, m, mixed_nbd=True)
    rows_o = []
    so = po.run(po.Engine("oracle"), p,
                snapshot=lambda k, s: rows_o.append((int(s.isave[29]), int(s.isave[33]), int(s.isave[32]),
                                                     int(s.isave[37]), int(s.isave[27]), float(s.f[0])))
                if s.task_s.startswith("NEW_X") else None)
    sol = la.DeviceSolver(n, m)
    xs = [torch.from_numpy(p.x0.copy()).cuda(), torch.zeros(n, dtype=torch.float64, device="cuda")]
    gs = [torch.zeros_like(xs[0]), torch.zeros_like(xs[0])]
    l, u = torch.from_numpy(p.l).cuda(), torch.from_numpy(p.u).cuda()
    nbd = torch.from_numpy(p.nbd.astype(np.int32)).cuda()
    rows_g, cur = [], 0
    for _ in range(100000):
        if pp:
            t, cur = sol.setulb_pp(xs, l, u, nbd, gs, 0.0, 0.0)
        else:
            t = sol.setulb(xs[0], l, u, nbd, gs[0], 0.0, 0.0)
        if t.startswith("FG"):
            sol.f[0] = sol.objective(0, xs[cur], gs[cur])
        elif t.startswith("NEW_X"):
            rows_g.append((int(sol.isave[29]), int(sol.isave[33]), int(sol.isave[32]), int(sol.isave[37]),
                           int(sol.isave[27]), float(sol.f[0])))
        else:
            break
    sol.close()
    assert t == so.task_s and t.startswith("CONVERGENCE")
    assert max(r[4] for r in rows_o) == m
    k = 0
    while k < min(len(rows_o), len(rows_g)) and rows_o[k][:5] == rows_g[k][:5]:
        k += 1
    assert k >= 60, (k, rows_o[k - 1:k + 1], rows_g[k - 1:k + 1])
    for a, b in zip(rows_g, rows_o):
        assert a[5] == pytest.approx(b[5], rel=1e-10)


def test_headline_config_n1e8_fp64_anchors(env):
    """The workload bench.py times (BASELINE.json metric: n = 1e8, m = 10, fp64, on-device
    objective) at full size, 14 iterations -- through the first full-sort walk, the filling of
    the memory and into the steady state -- against the rows the REAL reference printed for this very
    run (tests/golden/quad_n1e8_m10_ref_rows.json: -fdefault-integer-8 build on the GPU box's host,
    116 s for iteration 1, 11 s per iteration after): iteration, nfg, nseg, nfree of every iteration
    exactly (nseg(it1) = 97,671,921, nfree(it2) = 49,999,496, the 77,315 / 36,374 / 74,057-segment
    walks of iterations 4-6, the second line-search trial of iteration 14), f to 1e-9, |proj g| to
    1e-7; the two-pass iteration (closed-form W'Z r) must be the path taken once pairs are stored;
    and the sums over 1e8 rows must be reproducible bit for bit from one run to the next
    (fixed-order reductions, no atomics)."""
    import json
    torch, la = env["torch"], env["la"]
    n, m, iters = 100_000_000, 10, 14
    free_b, _tot = torch.cuda.mem_get_info()
    if free_b < 40 * (1 << 30):
        pytest.skip("needs ~30 GB of HBM")

    def run(timed_path=False):
        """timed_path: exactly what bench.py times -- lbfgsb_hip_setulb_dev_pp (ping-pong iterate buffers) on a
        context with LBFGSB_F_DEFER_LNSRCH, the objective deferred (its value rides with the next call's fetch)"""
        sol = la.DeviceSolver(n, m, same_stream_objective=timed_path, defer_lnsrch=timed_path)
        x = torch.zeros(n, dtype=torch.float64, device="cuda")
        g = torch.zeros_like(x)
        xs, gs = ([x, torch.empty_like(x)], [g, torch.empty_like(g)]) if timed_path else ([x], [g])
        l, u = torch.full_like(x, -1.0), torch.full_like(x, 1.0)
        nbd = torch.full((n,), 2, dtype=torch.int32, device="cuda")
        rows = []
        cur = 0
        while True:
            if timed_path:
                t, cur = sol.setulb_pp(xs, l, u, nbd, gs, 0.0, 0.0)
            else:
                t = sol.setulb(x, l, u, nbd, g, 0.0, 0.0)
            if t.startswith("FG"):
                if timed_path:
                    sol.objective(0, xs[cur], gs[cur], deferred=True)
                else:
                    sol.f[0] = sol.objective(0, x, g)
            elif t.startswith("NEW_X"):
                rows.append((int(sol.isave[29]), int(sol.isave[33]), int(sol.isave[32]), int(sol.isave[37]),
                             float(sol.f[0]), float(sol.dsave[12])))
                if sol.isave[29] >= iters:
                    break
            else:
                break
        counts = sol.path_counts()
        deferred = sol.defer_stats()
        sol.close()
        del x, g, l, u, nbd, xs, gs
        torch.cuda.empty_cache()
        return rows, counts, deferred
    rows, (closed_steps, three_steps, _), _d = run()
    assert len(rows) == iters
    assert rows[0][2] == 97_671_921, rows[0]
    assert rows[1][3] == 49_999_496, rows[1]
    assert all(b[4] < a[4] for a, b in zip(rows, rows[1:]))
    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                      "quad_n1e8_m10_ref_rows.json")))
    assert ref["n"] == n and ref["m"] == m
    for got, want in zip(rows, ref["rows"]):
        assert got[:4] == (want["iter"], want["nfg"], want["nseg"], want["nfree"]), (got, want)
        assert got[4] == pytest.approx(want["f"], rel=1e-9), (got, want)
        # (once a line search has interpolated -- nfg > iter + 1, iteration 14 here -- its step comes
        #  from DIFFERENCES of f: f ~ 4.2e8 falls by ~1.6e3 per iteration, so the 1e-11 relative
        #  difference between the caller's two ways of summing 1e8 terms of f is amplified by
        #  f / delta f ~ 2.6e5 into the step, hence into x and |proj g|)
        assert got[5] == pytest.approx(want["sbgnrm"], rel=1e-7 if want["nfg"] == want["iter"] + 1 else 1e-5), \
            (got, want)
    assert closed_steps >= 8, (closed_steps, three_steps)
    rows2, _, _d = run()
    assert rows2 == rows       # bit for bit, f and |proj g| included
    # THE TIMED PATH at size: ping-pong entry + LBFGSB_F_DEFER_LNSRCH + deferred f (what bench.py drives).  Its
    # contract is "every NEW_X return bit for bit the default's": integers, |proj g| (same kernels, same order)
    # exactly; f differs from the classic run only by WHERE the objective's partial sums are completed (the
    # deferred value is reduced by the next pass's finalize instead of its own) -- still within 1e-12
    rows3, (closed3, _t3, _h3), (ndef, nredo) = run(timed_path=True)
    assert len(rows3) == iters and ndef >= iters - 2 and closed3 >= 8, (ndef, nredo, closed3)
    for got, want in zip(rows3, rows):
        assert got[:4] == want[:4], (got, want)
        assert got[4] == pytest.approx(want[4], rel=1e-12) and got[5] == pytest.approx(want[5], rel=1e-9), (got, want)
    for got, want in zip(rows3, ref["rows"]):
        assert got[:4] == (want["iter"], want["nfg"], want["nseg"], want["nfree"]), (got, want)
        assert got[4] == pytest.approx(want["f"], rel=1e-9), (got, want)


def test_unconstrained_problem_takes_the_two_pass_iteration(env):
    """nbd = 0 everywhere (mainlb :607-611: no Cauchy search, z = x): the update pass is run for
    unconstrained problems too -- its p = W'd over all rows IS W'Z r, the pair stays pending and
    is committed by the subspace pass -- so the iteration is the same two passes over W.
    Trajectory against the oracle: integer columns exactly, f to 1e-10, the closed form taken."""
    po, torch, la = env["po"], env["torch"], env["la"]
    n, m, iters = 200_003, 10, 25
    base = po.problem_quadratic(n, m)
    p = po.Problem("quad_unconstrained", n, m, base.x0, base.l, base.u, np.zeros(n, np.int32), 0.0, 0.0,
                   base.fg, np.float64)
    rows_o = []
    po.run(po.Engine("oracle"), p, max_iter=iters,
           snapshot=lambda k, s: rows_o.append((int(s.isave[29]), int(s.isave[33]), int(s.isave[32]),
                                                int(s.isave[37]), float(s.f[0]))) if s.task_s.startswith("NEW_X") else None)
    sol = la.DeviceSolver(n, m)
    x = torch.from_numpy(p.x0.copy()).cuda()
    g = torch.zeros_like(x)
    l, u = torch.from_numpy(p.l).cuda(), torch.from_numpy(p.u).cuda()
    nbd = torch.zeros(n, dtype=torch.int32, device="cuda")
    rows_g = []
    while True:
        t = sol.setulb(x, l, u, nbd, g, 0.0, 0.0)
        if t.startswith("FG"):
            sol.f[0] = sol.objective(0, x, g)
        elif t.startswith("NEW_X"):
            rows_g.append((int(sol.isave[29]), int(sol.isave[33]), int(sol.isave[32]), int(sol.isave[37]),
                           float(sol.f[0])))
            if sol.isave[29] >= iters:
                break
        else:
            break
    closed_steps, three_steps, _ = sol.path_counts()
    sol.close()
    assert len(rows_g) == len(rows_o)
    for a, b in zip(rows_g, rows_o):
        assert a[:4] == b[:4], (a, b)
        assert a[4] == pytest.approx(b[4], rel=1e-10)
    assert closed_steps >= len(rows_o) - 3, (closed_steps, three_steps)


def test_few_free_variables_still_two_passes(env):
    """A solution with ~1 % of the variables free (the minimiser of the other 99 % lies far outside
    the box): the closed form for W'Z r is guarded by how much of each stored s_i lives on the free
    rows -- not by how many rows are free -- and variables that sit at a bound do not move, so the
    iteration stays at two passes.  Trajectory against the oracle: integers exactly, f to 1e-10."""
    po, torch, la = env["po"], env["torch"], env["la"]
    n, m, iters = 120_000, 8, 30
    i = np.arange(1, n + 1)
    a = 1.0 + 99.0 * ((7919 * i) % 10007) / 10006.0
    c = np.where(i % 100 == 0, -0.5 + ((104729 * i) % 100003) / 100002.0, 5.0 + (i % 7))

    def fg(x, g):
        d = x - c
        g[:] = a * d
        return float(0.5 * np.sum(a * d * d))
    p = po.Problem("mostly_active", n, m, np.zeros(n), -np.ones(n), np.ones(n), np.full(n, 2, np.int32),
                   0.0, 0.0, fg, np.float64)
    rows_o = []
    po.run(po.Engine("oracle"), p, max_iter=iters,
           snapshot=lambda k, s: rows_o.append((int(s.isave[29]), int(s.isave[33]), int(s.isave[32]),
                                                int(s.isave[37]), float(s.f[0]))) if s.task_s.startswith("NEW_X") else None)
    assert rows_o[-1][3] <= n // 50            # few free variables indeed
    sol = la.DeviceSolver(n, m)
    x = torch.zeros(n, dtype=torch.float64, device="cuda")
    g = torch.zeros_like(x)
    l, u = torch.full_like(x, -1.0), torch.full_like(x, 1.0)
    nbd = torch.full((n,), 2, dtype=torch.int32, device="cuda")
    rows_g = []
    while True:
        t = sol.setulb(x, l, u, nbd, g, 0.0, 0.0)
        if t.startswith("FG"):
            xh = x.cpu().numpy()
            gh = np.empty_like(xh)
            sol.f[0] = fg(xh, gh)
            g.copy_(torch.from_numpy(gh))
        elif t.startswith("NEW_X"):
            rows_g.append((int(sol.isave[29]), int(sol.isave[33]), int(sol.isave[32]), int(sol.isave[37]),
                           float(sol.f[0])))
            if sol.isave[29] >= iters:
                break
        else:
            break
    closed_steps, three_steps, _ = sol.path_counts()
    sol.close()
    assert len(rows_g) == len(rows_o)
    for ra, rb in zip(rows_g, rows_o):
        assert ra[:4] == rb[:4], (ra, rb)
        assert ra[4] == pytest.approx(rb[4], rel=1e-10)
    assert closed_steps >= len(rows_o) // 2, (closed_steps, three_steps)


def test_full_size_rosenbrock_n1e6_against_oracle(env):
    """BASELINE.json configs[2] shape (extended Rosenbrock with box bounds, driver3 formulas) at
    n = 1e6, m = 10, on-device objective.  Iteration 1 fixes 999,999 variables in two massive
    TIE groups of breakpoints (x0 is uniform): nseg = 1,000,000, nfree = 1.  Anchors measured on
    the reference: it1 f = 9.9898699732555956E+07 (nfg 5),
    it2 f = 6.0004592590896189E+06, it10 f = 2.8578393725511319E+01."""
    po, torch, la = env["po"], env["torch"], env["la"]
    n, m, iters = 1_000_000, 10, 14
    p = po.problem_rosenbrock(n, m, 0.0, 0.0)
    rows_o = []
    po.run(po.Engine("oracle"), p, max_iter=iters,
           snapshot=lambda k, s: rows_o.append((int(s.isave[29]), int(s.isave[33]), int(s.isave[32]),
                                                int(s.isave[37]), float(s.f[0]))) if s.task_s.startswith("NEW_X") else None)
    assert rows_o[0][:4] == (1, 5, 1000000, 1)
    assert rows_o[0][4] == pytest.approx(9.9898699732555956e07, rel=1e-13)
    assert rows_o[1][4] == pytest.approx(6.0004592590896189e06, rel=1e-13)
    assert rows_o[9][4] == pytest.approx(2.8578393725511319e01, rel=1e-12)
    sol = la.DeviceSolver(n, m)
    x = torch.full((n,), 3.0, dtype=torch.float64, device="cuda")
    g = torch.zeros_like(x)
    l = torch.from_numpy(p.l).cuda()
    u = torch.from_numpy(p.u).cuda()
    nbd = torch.full((n,), 2, dtype=torch.int32, device="cuda")
    rows_g = []
    while True:
        t = sol.setulb(x, l, u, nbd, g, 0.0, 0.0)
        if t.startswith("FG"):
            sol.f[0] = sol.objective(1, x, g)
        elif t.startswith("NEW_X"):
            rows_g.append((int(sol.isave[29]), int(sol.isave[33]), int(sol.isave[32]),
                           int(sol.isave[37]), float(sol.f[0])))
            if sol.isave[29] >= iters:
                break
        else:
            break
    sol.close()
    assert len(rows_g) == iters
    for a, b in zip(rows_g, rows_o):
        assert a[:4] == b[:4], (a, b)                 # iter, nfg, nseg, nfree
        assert a[4] == pytest.approx(b[4], rel=1e-8)


def test_config3_rosenbrock_n1e7_against_reference_anchors(env):
    """BASELINE.json configs[2] at its stated size: extended Rosenbrock with box bounds (driver3
    formulas), n = 1e7, m = 10, fp64, on-device objective -- against per-iteration anchors that the
    REAL reference produced in the build container (tests/golden/rosenbrock_n1e7_anchors.json,
    made by tests/golden/make_anchors_n1e7.py).  Iteration 1 fixes 9,999,999 variables in two
    tie groups; iterations 10 -> 11, 12 -> 13 and 15 -> 16 move ~5e6 variables between the free and
    the active set (SURVEY.md appendix C's stress pattern: formk's from-scratch Gram path)."""
    import json
    po, torch, la = env["po"], env["torch"], env["la"]
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                       "rosenbrock_n1e7_anchors.json")))
    n, m = gold["n"], gold["m"]
    rows_o = [(r["iter"], r["nfg"], r["nseg"], r["nfree"], r["f"]) for r in gold["rows"]]
    iters = len(rows_o)
    assert n == 10_000_000 and rows_o[0][2] == n and rows_o[0][3] == 1
    assert any(abs(a[3] - b[3]) > 4_000_000 for a, b in zip(rows_o, rows_o[1:]))
    sol = la.DeviceSolver(n, m)
    x = torch.full((n,), 3.0, dtype=torch.float64, device="cuda")
    g = torch.zeros_like(x)
    l = torch.empty_like(x)
    l[0::2] = 1.0
    l[1::2] = -100.0
    u = torch.full_like(x, 100.0)
    nbd = torch.full((n,), 2, dtype=torch.int32, device="cuda")
    rows_g = []
    while True:
        t = sol.setulb(x, l, u, nbd, g, 0.0, 0.0)
        if t.startswith("FG"):
            sol.f[0] = sol.objective(1, x, g)
        elif t.startswith("NEW_X"):
            rows_g.append((int(sol.isave[29]), int(sol.isave[33]), int(sol.isave[32]), int(sol.isave[37]),
                           float(sol.f[0])))
            if sol.isave[29] >= iters:
                break
        else:
            break
    sol.close()
    del x, g, l, u, nbd
    torch.cuda.empty_cache()
    assert len(rows_g) == iters
    for a, b in zip(rows_g, rows_o):
        assert a[:4] == b[:4], (a, b)                 # iter, nfg, nseg, nfree
        assert a[4] == pytest.approx(b[4], rel=1e-7)


def _random_box_rosenbrock(po, seed):
    """Extended Rosenbrock with a random start, random boxes around it and all four bound
    types: small problems that reach the rarely taken branches of the reference."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(10, 120))
    m = int(rng.integers(2, 8))
    p = po.problem_rosenbrock(n, m, 0.0, 0.0)
    x0 = rng.uniform(-2.5, 2.5, n)
    w = rng.uniform(0.05, 3.0, n)
    p.x0[:] = x0
    p.l[:] = x0 - w * rng.random(n)
    p.u[:] = x0 + w * rng.random(n)
    p.nbd[:] = rng.integers(0, 4, n)
    return p


@pytest.mark.parametrize("seed", [5003, 5021, 5028, 5006, 5007])
def test_rare_branches_one_step(env, seed):
    """The subsm backtracking branch ('Positive dir derivative in projection', reference
    :2830-2879: alpha with arg-min, snap to the bound) and the 'refresh the lbfgs memory'
    branches of mainlb: the oracle's branch counters locate the calls that take them; those
    calls (and their neighbours) are replayed on the GPU from the identical state."""
    import ctypes as C
    po = env["po"]
    p = _random_box_rosenbrock(po, seed)
    eng = po.Engine("oracle")
    cnt = (C.c_long * 4).in_dll(eng.lib, "lbo_branch_count")
    cnt[0] = cnt[1] = 0
    snaps, marks = [], []

    def snap(k, s):
        snaps.append(s.copy())
        marks.append((cnt[0], cnt[1]))
    po.run(eng, p, max_iter=150, snapshot=snap)
    hits = [k for k in range(1, len(marks)) if marks[k] != marks[k - 1]]
    assert hits, "this seed is expected to reach a rare branch"
    tested = 0
    for kb in hits[:2]:
        for k in range(max(1, kb - 1), min(len(snaps), kb + 2)):
            t = snaps[k - 1].task_s
            if not (t.startswith("FG") or t.startswith("NEW_X")):
                continue
            _, out = gpu_one_call(env, p, snaps[k - 1])
            compare_states(out, snaps[k], p.n, p.m, po, rtol=1e-9)
            tested += 1
    assert tested >= 2


def test_minimize_wrapper_matches_reverse_communication(env):
    """lbfgsb_hip_minimize (the wrapper the reference lists as @todo, src/lbfgsb.f90:36-37) runs
    the same loop as driver2: same final state as driving setulb by hand; built-in objective and
    a Python callback; the iteration cap ends with a 'STOP' task like the drivers do."""
    po, torch, la = env["po"], env["torch"], env["la"]
    n, m = 5003, 6
    p = po.problem_quadratic(n, m, mixed_nbd=True)
    ref = po.run(po.Engine("oracle"), p)
    assert ref.task_s.startswith("CONVERGENCE")

    def tensors():
        return (torch.from_numpy(p.x0.copy()).cuda(), torch.from_numpy(p.l).cuda(),
                torch.from_numpy(p.u).cuda(), torch.from_numpy(p.nbd.astype(np.int32)).cuda(),
                torch.zeros(n, dtype=torch.float64, device="cuda"))
    # the same run driven by hand through the reverse-communication entry
    hand = la.DeviceSolver(n, m)
    xh, lh, uh, nbdh, gh = tensors()
    while True:
        th = hand.setulb(xh, lh, uh, nbdh, gh, 0.0, 0.0)
        if th.startswith("FG"):
            hand.f[0] = hand.objective(0, xh, gh)
        elif not th.startswith("NEW_X"):
            break
    sol = la.DeviceSolver(n, m)
    x, l, u, nbd, g = tensors()
    t = sol.minimize(x, l, u, nbd, g, builtin=0, factr=0.0, pgtol=0.0)
    # the wrapper IS the hand-driven loop: same task, same counters, same point
    assert t == th
    assert int(sol.isave[29]) == int(hand.isave[29]) and int(sol.isave[33]) == int(hand.isave[33])
    assert float(sol.f[0]) == pytest.approx(float(hand.f[0]), rel=1e-13)
    assert np.max(np.abs(x.cpu().numpy() - xh.cpu().numpy())) <= 1e-12
    hand.close()
    # ... and the oracle's run: with factr = 0 the LAST iterations act on the rounding noise of f (the
    # stop test is "f did not decrease"), so how many of them there are depends on the order of the sums;
    # the point they stop at does not
    assert t == ref.task_s
    assert abs(int(sol.isave[29]) - int(ref.isave[29])) <= 4 and abs(int(sol.isave[33]) - int(ref.isave[33])) <= 6
    assert float(sol.f[0]) == pytest.approx(float(ref.f[0]), rel=1e-10)
    # (f is flat to 1e-10 around the minimiser; x is where those last iterations left it)
    assert np.max(np.abs(x.cpu().numpy() - ref.x)) <= 1e-6
    sol.close()
    # Python callback objective (device pointers) + iteration cap
    sol = la.DeviceSolver(n, m)
    x, l, u, nbd, g = tensors()
    calls = []

    def fg(xp, gp):
        assert xp == x.data_ptr() and gp == g.data_ptr()
        calls.append(1)
        return sol.objective(0, x, g)
    t = sol.minimize(x, l, u, nbd, g, fg=fg, factr=0.0, pgtol=0.0, max_iter=7)
    assert t.startswith("STOP: MAXIMUM NUMBER OF ITERATIONS") and int(sol.isave[29]) == 7
    assert len(calls) == int(sol.isave[33])
    sol.close()


def test_run_to_convergence_with_the_drivers_tolerances_n1e6(env):
    """Time to SOLUTION, not just iterations: the bounded quadratic at n = 1e6, m = 10 with driver1's own
    stopping tolerances (test/driver1.f90:212-213: factr = 1e7, pgtol = 1e-5) run to its CONVERGENCE message
    (src/lbfgsb.f90:795-810) -- through lbfgsb_hip_minimize with the built-in objective, and through the timed
    path's entry (ping-pong + deferred set-up) -- against the LIVE reference (oracle/_ref; the C oracle if it is
    not built) on the same problem: same message, same iteration and evaluation counts (a stop by relative
    reduction at factr = 1e7 is decided four decades above rounding noise: an exact anchor), f to 1e-10."""
    po, torch, la = env["po"], env["torch"], env["la"]
    n, m = 1_000_000, 10
    base = po.problem_quadratic(n, m)
    p = po.Problem("quadratic_conv", n, m, base.x0, base.l, base.u, base.nbd, 1.0e7, 1.0e-5, base.fg, np.float64)
    eng = po.Engine("ref") if po.Engine.available("ref") else po.Engine("oracle")
    ref = po.run(eng, p)
    assert ref.task_s.startswith("CONVERGENCE: REL_REDUCTION_OF_F_<=_FACTR*EPSMCH"), ref.task_s
    it_ref, nfg_ref, f_ref = int(ref.isave[29]), int(ref.isave[33]), float(ref.f[0])
    assert 10 <= it_ref <= 200

    def tensors():
        x = torch.zeros(n, dtype=torch.float64, device="cuda")
        return (x, torch.full_like(x, -1.0), torch.full_like(x, 1.0),
                torch.full((n,), 2, dtype=torch.int32, device="cuda"), torch.zeros_like(x))
    sol = la.DeviceSolver(n, m)
    x, l, u, nbd, g = tensors()
    t = sol.minimize(x, l, u, nbd, g, builtin=0, factr=p.factr, pgtol=p.pgtol)
    assert t == ref.task_s
    assert (int(sol.isave[29]), int(sol.isave[33])) == (it_ref, nfg_ref)
    assert float(sol.f[0]) == pytest.approx(f_ref, rel=1e-10)
    assert np.max(np.abs(x.cpu().numpy() - ref.x)) <= 1e-7
    sol.close()
    # the timed path's entry and flags, by hand
    sol = la.DeviceSolver(n, m, same_stream_objective=True, defer_lnsrch=True)
    x, l, u, nbd, g = tensors()
    xs, gs = [x, torch.empty_like(x)], [g, torch.empty_like(g)]
    cur = 0
    while True:
        t, cur = sol.setulb_pp(xs, l, u, nbd, gs, p.factr, p.pgtol)
        if t.startswith("FG"):
            sol.objective(0, xs[cur], gs[cur], deferred=True)
        elif not t.startswith("NEW_X"):
            break
    sol.sync()
    assert t == ref.task_s
    assert (int(sol.isave[29]), int(sol.isave[33])) == (it_ref, nfg_ref)
    assert float(sol.f[0]) == pytest.approx(f_ref, rel=1e-10)
    assert np.max(np.abs(xs[cur].cpu().numpy() - ref.x)) <= 1e-7
    sol.close()


@pytest.mark.parametrize("kind", ["quadratic", "quadmix", "rosenbrock"])
def test_parallel_gcp_opt_in(env, kind):
    """LBFGSB_F_PARALLEL_GCP (col = 0 case): the closed-form GCP
    t* = 1/theta replaces the ordered walk of src/lbfgsb.f90:1378-1497 when no pair is stored.
    Equal in exact arithmetic; the reference's f1/f2 recurrence has its own rounding noise, so
    the tolerance here is the opt-in mode's documented one: nseg within 2 of the oracle's,
    f to 1e-9, and no host sorting/walking at all (no full sort, a handful of syncs)."""
    po, torch, la = env["po"], env["torch"], env["la"]
    n, m, iters = 200_000, 10, 3
    if kind == "rosenbrock":
        p = po.problem_rosenbrock(n, m, factr=0.0, pgtol=0.0)
    else:
        p = po.problem_quadratic(n, m, mixed_nbd=(kind == "quadmix"))
    rows_o = []
    po.run(po.Engine("oracle"), p, max_iter=iters,
           snapshot=lambda k, s: rows_o.append((int(s.isave[29]), int(s.isave[33]), int(s.isave[32]),
                                                int(s.isave[37]), float(s.f[0]))) if s.task_s.startswith("NEW_X") else None)
    sol = la.DeviceSolver(n, m, parallel_gcp=True)
    x = torch.from_numpy(p.x0.copy()).cuda()
    g = torch.zeros_like(x)
    l, u = torch.from_numpy(p.l).cuda(), torch.from_numpy(p.u).cuda()
    nbd = torch.from_numpy(p.nbd.astype(np.int32)).cuda()
    rows_g = []
    st1 = None
    while True:
        t = sol.setulb(x, l, u, nbd, g, 0.0, 0.0)
        if t.startswith("FG"):
            sol.f[0] = sol.objective(0 if kind != "rosenbrock" else 1, x, g)
            if st1 is None:
                pass
        elif t.startswith("NEW_X"):
            if st1 is None:
                st1 = sol.stats()
            rows_g.append((int(sol.isave[29]), int(sol.isave[33]), int(sol.isave[32]),
                           int(sol.isave[37]), float(sol.f[0])))
            if sol.isave[29] >= iters:
                break
        else:
            break
    sol.close()
    assert rows_o[0][2] > n // 4            # iteration 1 really is the nseg ~ n case
    assert st1["cauchy_fullsorts"] == 0      # and it was done without sorting or walking
    # index work of the closed form, stated independently: with no pair stored (theta = 1) the
    # walk crosses exactly the breakpoints t_j <= 1/theta (src/lbfgsb.f90:1270-1330 for t_j, the
    # exit test "dtm < dt" of :1416 for the <=): nseg of iteration 1 must EQUAL 1 + their number
    g0 = np.empty(n)
    p.fg(p.x0, g0)
    neg = -g0
    tl, tu = p.x0 - p.l, p.u - p.x0
    has_l, has_u = (p.nbd == 1) | (p.nbd == 2), (p.nbd == 2) | (p.nbd == 3)
    stuck = (has_l & (tl <= 0) & (neg <= 0)) | (has_u & ~(has_l & (tl <= 0)) & (tu <= 0) & (neg >= 0))
    with np.errstate(divide="ignore", invalid="ignore"):
        tb = np.where(has_l & (neg < 0), tl / -neg, np.where(has_u & (neg > 0), tu / neg, np.inf))
    tb[stuck | (neg == 0)] = np.inf
    crossed = int(np.sum(tb <= 1.0))
    assert crossed < n                       # (else the last crossing counts no segment, :1436)
    assert rows_g[0][2] == 1 + crossed, (rows_g[0], crossed)
    assert len(rows_g) == len(rows_o) == iters
    for a, b in zip(rows_g, rows_o):
        assert a[:2] == b[:2], (a, b)
        assert abs(a[2] - b[2]) <= 2 and abs(a[3] - b[3]) <= 2, (a, b)
        assert a[4] == pytest.approx(b[4], rel=1e-9)


@pytest.mark.parametrize("handover", [False, True], ids=["", "handover"])
@pytest.mark.parametrize("n,m,mixed,min_agree", [(1000, 10, False, 72), (4096, 10, True, 80),
                                                 (100003, 5, False, 60)])
def test_device_path_to_convergence_against_oracle(env, n, m, mixed, min_agree, handover):
    """The production path (device pointers, speculative update pass, pending pair, functional
    Cauchy point, on-device objective) run to CONVERGENCE with factr = pgtol = 0: the integer
    columns (iteration, nfg, nseg, nfree) equal the oracle's for at least `min_agree` iterations
    (all 72 of the n = 1000 fixture; beyond that the stop test acts on rounding noise), f agrees
    to 1e-11 throughout, same final message.  `handover`: with the option spec_capture = 1 the update
    pass also hands the next walk's first breakpoints over (off by default: DESIGN.md section 4)."""
    po, torch, la = env["po"], env["torch"], env["la"]
    p = po.problem_quadratic(n, m, mixed_nbd=mixed)
    rows_o = []
    so = po.run(po.Engine("oracle"), p,
                snapshot=lambda k, s: rows_o.append((int(s.isave[29]), int(s.isave[33]), int(s.isave[32]),
                                                     int(s.isave[37]), float(s.f[0])))
                if s.task_s.startswith("NEW_X") else None)
    sol = la.DeviceSolver(n, m, options={"spec_capture": 1} if handover else None)
    x = torch.from_numpy(p.x0.copy()).cuda()
    g = torch.zeros_like(x)
    l, u = torch.from_numpy(p.l).cuda(), torch.from_numpy(p.u).cuda()
    nbd = torch.from_numpy(p.nbd.astype(np.int32)).cuda()
    rows_g = []
    for _ in range(100000):
        t = sol.setulb(x, l, u, nbd, g, 0.0, 0.0)
        if t.startswith("FG"):
            sol.f[0] = sol.objective(0, x, g)
        elif t.startswith("NEW_X"):
            rows_g.append((int(sol.isave[29]), int(sol.isave[33]), int(sol.isave[32]),
                           int(sol.isave[37]), float(sol.f[0])))
        else:
            break
    handed = sol.path_counts()[2]
    sol.close()
    assert handed == 0 if not handover else handed >= 0
    assert t == so.task_s and t.startswith("CONVERGENCE")
    k = 0
    while k < min(len(rows_o), len(rows_g)) and rows_o[k][:4] == rows_g[k][:4]:
        k += 1
    assert k >= min_agree, (k, rows_o[k - 1:k + 1], rows_g[k - 1:k + 1])
    for a, b in zip(rows_g, rows_o):
        assert a[4] == pytest.approx(b[4], rel=1e-11)
    assert float(sol.f[0]) == pytest.approx(float(so.f[0]), rel=1e-11)


@pytest.mark.parametrize("n,m,mixed", [(1000, 10, False), (4096, 10, True), (100003, 5, False)])
def test_runs_to_convergence_under_the_replay_bar(env, n, m, mixed):
    """The same three runs to CONVERGENCE (factr = pgtol = 0: the last iterations act on rounding noise) call by
    call beside the oracle, both device-pointer entries: where a run leaves the oracle's trajectory -- the
    `min_agree` of the test above -- ONE oracle call from the run's own previous state must reproduce the call
    (tests/test_gpu_fuzz.py: drive_with_replay), and the final f still agrees to 1e-7."""
    from test_gpu_fuzz import drive_with_replay
    po = env["po"]
    p = po.problem_quadratic(n, m, mixed_nbd=mixed)
    for pp in (False, True):
        split, ncalls = drive_with_replay(po, p, 10 ** 9, pp=pp)
        assert ncalls > 100 and (split is None or split > 50), (split, ncalls)


def test_parallel_gcp_with_pairs_stored(env):
    """LBFGSB_F_PARALLEL_GCP when pairs are stored (col > 0): a two-scale
    separable quadratic whose 2nd and 6th iterations each cross ~91 000 breakpoints with col = 1
    and col = m = 5.  The walk is replaced by a full sort + prefix scans on the device
    (parallel_gcp in solver.hip); nseg / nfree must agree with the oracle's sequential walk to
    within 2, f to 1e-9, and the path must really have been taken (two full sorts; the exact
    replay needs none on this problem)."""
    po, torch, la = env["po"], env["torch"], env["la"]
    n, m, iters = 200_000, 5, 8
    rng = np.random.default_rng(7)
    a = 1.0 + 99.0 * rng.random(n)
    a[n // 2:] *= 1e-4
    c = rng.choice([-1.0, 1.0], n) * 5.0 * (1.0 + rng.random(n))
    eps = 1e-3

    def fg(x, g):
        d = x - c
        g[:] = eps * a * d
        return float(0.5 * eps * np.sum(a * d * d))
    p = po.Problem("two_scale", n, m, np.zeros(n), -np.ones(n), np.ones(n), np.full(n, 2, np.int32),
                   0.0, 0.0, fg, np.float64)
    rows_o = []
    po.run(po.Engine("oracle"), p, max_iter=iters,
           snapshot=lambda k, s: rows_o.append((int(s.isave[29]), int(s.isave[33]), int(s.isave[32]),
                                                int(s.isave[37]), int(s.isave[27]), float(s.f[0])))
           if s.task_s.startswith("NEW_X") else None)
    assert sum(1 for r in rows_o if r[2] > 50_000 and r[4] > 0) >= 2      # long walks with col > 0
    sol = la.DeviceSolver(n, m, parallel_gcp=True)
    x = torch.zeros(n, dtype=torch.float64, device="cuda")
    g = torch.zeros_like(x)
    l, u = torch.full_like(x, -1.0), torch.full_like(x, 1.0)
    nbd = torch.full((n,), 2, dtype=torch.int32, device="cuda")
    rows_g = []
    while True:
        t = sol.setulb(x, l, u, nbd, g, 0.0, 0.0)
        if t.startswith("FG"):
            xh = x.cpu().numpy()
            gh = np.empty_like(xh)
            sol.f[0] = fg(xh, gh)
            g.copy_(torch.from_numpy(gh))
        elif t.startswith("NEW_X"):
            rows_g.append((int(sol.isave[29]), int(sol.isave[33]), int(sol.isave[32]),
                           int(sol.isave[37]), int(sol.isave[27]), float(sol.f[0])))
            if sol.isave[29] >= iters:
                break
        else:
            break
    st = sol.stats()
    sol.close()
    assert len(rows_g) == len(rows_o)
    for a_, b_ in zip(rows_g, rows_o):
        assert a_[:2] == b_[:2] and a_[4] == b_[4], (a_, b_)
        assert abs(a_[2] - b_[2]) <= 2 and abs(a_[3] - b_[3]) <= 2, (a_, b_)
        assert a_[5] == pytest.approx(b_[5], rel=1e-9)
    assert st["cauchy_fullsorts"] >= 2


# ---------------------------------------------------------------------------------------------
# Equal breakpoints (src/lbfgsb.f90:1384-1403, hpsolb :2079-2157)
# ---------------------------------------------------------------------------------------------
def _craft_tie_split(po, want_non_prefix=True):
    """A one-step input whose Cauchy walk ENDS INSIDE a group of equal breakpoints, found with
    the oracle: at a NEW_X state of a quadratic run, K interior free variables get the same
    distance to their lower bound (delta, exactly) and the same gradient entry (gamma), hence
    bit-identical breakpoints t* = delta / gamma, while their rows of W differ -- the jumps of
    f' at t* differ per variable and f' changes sign part-way through the group.
    -> (problem, state, group, oracle's next state)."""
    n, m = 600, 5
    p = po.problem_quadratic(n, m)
    eng = po.Engine("oracle")
    snaps = []
    po.run(eng, p, max_calls=60, snapshot=lambda k, s: snaps.append(s.copy()))
    off = po.wa_offsets(n, m)
    rng = np.random.default_rng(1)

    def step(prob, s):
        s = s.copy()
        po.call(eng, prob, s)
        return s
    for k, s in enumerate(snaps):
        if not s.task_s.startswith("NEW_X") or s.isave[27] < 2:
            continue
        nxt = step(p, s)
        x, d = s.x, -s.g
        o, ln = off["xp"]
        iw = nxt.iwa[n:2 * n]
        free = (iw == 0) & (np.abs(d) > 1e-8)
        if free.sum() < 50:
            continue
        ts = float(np.median((nxt.wa[o:o + ln][free] - x[free]) / d[free]))
        rest = float(np.sum(s.g[free] ** 2))
        interior = np.nonzero(free & (x > p.l + 0.3) & (x < p.u - 0.05))[0]
        for K in (8, 16):
            for _ in range(3):
                cand = np.sort(rng.choice(interior, K, replace=False))
                for mult in (0.25, 0.5, 1, 2, 4):
                    for c in (0.6, 0.9, 1.2):
                        gam = float(np.sqrt(rest * mult / K))
                        delta = float(np.round(c * ts * gam * 1024) / 1024)
                        if not 0 < delta <= 0.25:
                            continue
                        q = po.Problem(p.name, n, m, p.x0, p.l.copy(), p.u.copy(), p.nbd.copy(), p.factr,
                                       p.pgtol, p.fg, p.real)
                        s2 = s.copy()
                        if any(x[i] - (x[i] - delta) != delta for i in cand):
                            continue
                        q.l[cand] = x[cand] - delta
                        s2.g[cand] = gam
                        out = step(q, s2)
                        fixed = out.iwa[n:2 * n][cand] > 0
                        nfix = int(fixed.sum())
                        if not (0 < nfix < K and out.task_s.startswith("FG")):
                            continue
                        prefix = bool(np.all(fixed[:nfix]))
                        if want_non_prefix and prefix:
                            continue
                        return q, s2, cand, out
    raise AssertionError("no tie-split construction found")


def _one_call(env, p, s_in, mirror=True, **ctx_flags):
    po, torch, la = env["po"], env["torch"], env["la"]
    s = s_in.copy()
    sol = la.DeviceSolver(p.n, p.m, mirror_index=mirror, **ctx_flags)
    try:
        x, g = _dev(torch, s.x), _dev(torch, s.g)
        l, u, nbd = _dev(torch, p.l), _dev(torch, p.u), _dev(torch, p.nbd.astype(np.int32))
        sol.import_state(s.wa, s.iwa, s.isave)
        for name in ("task", "csave", "lsave", "isave", "dsave"):
            getattr(sol, name)[:] = getattr(s, name)
        sol.f[0] = s.f[0]
        sol.setulb(x, l, u, nbd, g, p.factr, p.pgtol)
        torch.cuda.synchronize()
        wa, iwa = sol.export_state()
        out = po.State(p.n, p.m, x.cpu().numpy(), g.cpu().numpy(), sol.f.copy(), wa, iwa, sol.task.copy(),
                       sol.csave.copy(), sol.lsave.copy(), sol.isave.copy(), sol.dsave.copy())
        return out, sol.tie_splits()
    finally:
        sol.close()


def test_walk_ending_inside_a_tie_group(env):
    """The case VERDICT r1 / r2 asked for: a walk that ends INSIDE a group of equal breakpoints,
    where the reference's heap order (hpsolb :2079) -- not the variable order -- decides which
    members are fixed (here a non-prefix subset of the group).
      * DEFAULT context: the call is detected (tie_splits == 1), the walk is replayed in the
        reference's own order and the whole state equals the oracle's as in every other one-step
        test (iwhere, Index, counters exactly);
      * LBFGSB_F_INDEX_TIES (opt-out): the generalized Cauchy POINT is still the reference's (every
        member of the group reaches its bound at t* whether it is labelled fixed or not: xp to
        1e-12), iwhere differs from the reference's only inside the group (a prefix of it)."""
    po = env["po"]
    q, s, group, want = _craft_tie_split(po)
    n, m = q.n, q.m
    fixed_ref = want.iwa[n:2 * n][group] > 0
    assert 0 < fixed_ref.sum() < len(group) and not np.all(fixed_ref[:fixed_ref.sum()])
    # default flags: the reference's order, full one-step parity
    got, splits = _one_call(env, q, s)
    assert splits == 1
    compare_states(got, want, n, m, po)
    assert np.array_equal(got.iwa[n:2 * n], want.iwa[n:2 * n])
    # the same through a production (non-mirror) context
    got_p, splits_p = _one_call(env, q, s, mirror=False)
    assert splits_p == 1
    assert np.array_equal(got_p.iwa[n:2 * n], want.iwa[n:2 * n])
    # opt-out: variable order
    got, splits = _one_call(env, q, s, index_ties=True)
    assert splits == 1
    off = po.wa_offsets(n, m)
    o, ln = off["xp"]
    assert np.max(np.abs(got.wa[o:o + ln] - want.wa[o:o + ln])) <= 1e-12
    diff = np.nonzero(got.iwa[n:2 * n] != want.iwa[n:2 * n])[0]
    assert set(diff.tolist()) <= set(group.tolist())
    fixed_got = got.iwa[n:2 * n][group] > 0
    assert np.all(fixed_got[:fixed_got.sum()])            # variable order: a prefix of the group


def test_tie_groups_of_identical_variables_trajectory(env):
    """Problems with EXACT symmetries (copies of the same variable) keep whole groups of equal
    breakpoints alive over many iterations.  A walk that ends inside such a group fixes the same
    NUMBER of copies in either order (identical jumps), so with LBFGSB_F_INDEX_TIES (variable
    order) the run and the reference stay equal in every scalar -- iteration, nfg, nseg, nfree, f --
    and in x up to a permutation inside the groups; the DEFAULT context replays such walks in the
    reference's order and x itself is equal."""
    po, torch, la = env["po"], env["torch"], env["la"]
    base, copies, m, iters = 257, 8, 5, 40
    n = base * copies
    b = np.arange(n) % base
    a = 1.0 + 99.0 * ((7919 * (b + 1)) % 10007) / 10006.0
    c = -2.0 + 4.0 * ((104729 * (b + 1)) % 100003) / 100002.0

    def fg(x, g):
        d = x - c
        g[:] = a * d
        return float(0.5 * np.sum(a * d * d))
    p = po.Problem("sym_quadratic", n, m, np.zeros(n), -np.ones(n), np.ones(n), np.full(n, 2, np.int32),
                   0.0, 0.0, fg, np.float64)
    rows_o, xs_o = [], []

    def snap(k, s):
        if s.task_s.startswith("NEW_X"):
            rows_o.append((int(s.isave[29]), int(s.isave[33]), int(s.isave[32]), int(s.isave[37]), float(s.f[0])))
            xs_o.append(s.x.copy())
    po.run(po.Engine("oracle"), p, max_iter=iters, snapshot=snap)

    def run_gpu(**flags):
        sol = la.DeviceSolver(n, m, **flags)
        x = torch.zeros(n, dtype=torch.float64, device="cuda")
        g = torch.zeros_like(x)
        l, u = torch.full_like(x, -1.0), torch.full_like(x, 1.0)
        nbd = torch.full((n,), 2, dtype=torch.int32, device="cuda")
        rows, xs = [], []
        while True:
            t = sol.setulb(x, l, u, nbd, g, 0.0, 0.0)
            if t.startswith("FG"):
                xh = x.cpu().numpy()
                gh = np.empty_like(xh)
                sol.f[0] = fg(xh, gh)
                g.copy_(torch.from_numpy(gh))
            elif t.startswith("NEW_X"):
                rows.append((int(sol.isave[29]), int(sol.isave[33]), int(sol.isave[32]), int(sol.isave[37]),
                             float(sol.f[0])))
                xs.append(x.cpu().numpy())
                if sol.isave[29] >= iters:
                    break
            else:
                break
        splits = sol.tie_splits()
        sol.close()
        return rows, xs, splits
    rows_g, xs_g, splits = run_gpu(index_ties=True)
    assert len(rows_g) == len(rows_o)
    for ra, rb in zip(rows_g, rows_o):
        assert ra[:4] == rb[:4], (ra, rb)
        assert ra[4] == pytest.approx(rb[4], rel=1e-10)
    for xa, xb in zip(xs_g, xs_o):
        ga = np.sort(xa.reshape(copies, base), axis=0)
        gb = np.sort(xb.reshape(copies, base), axis=0)
        assert np.max(np.abs(ga - gb)) <= 1e-9
    rows_e, xs_e, splits_e = run_gpu()
    assert [r[:4] for r in rows_e] == [r[:4] for r in rows_o]
    for xa, xb in zip(xs_e, xs_o):
        assert np.max(np.abs(xa - xb)) <= 1e-9
    print("tie splits on this trajectory: %d (index order), %d (default: replayed)" % (splits, splits_e))


def test_zz_factor_errors_are_explained_by_conditioning(env):
    """Runs last in this module: the one-step and production-path cases above compared wt (formt's
    Cholesky factor) and wn (formk's factored K) with loose absolute tolerances (1e-7, 1e-6); each
    comparison also required the error to be at most 50 x cond x (error of the factored matrix's
    inputs).  This reports the largest such amplification seen and insists the checks really ran."""
    assert COND_SEEN["wt"] > 0.0 and COND_SEEN["wn"] > 0.0
    print("largest error / (cond x input error): wt %.3g, wn %.3g" % (COND_SEEN["wt"], COND_SEEN["wn"]))
    assert COND_SEEN["wt"] <= 50.0 and COND_SEEN["wn"] <= 50.0

"""Routine-by-routine parity: every routine door of the C ABI (include/lbfgsb_hip.h "Routine doors",
SURVEY.md 8(b)(4)) against its twin in the oracle, on the same inputs.

The inputs of a routine come from the oracle's own trajectory: the state at a NEW_X return (wa, iwa,
isave, dsave in the reference's layout) and, from there, the reference's sequence of calls for the next
iteration -- matupd (+ formt), cauchy, freev, formk, cmprlb, subsm, lnsrlb (src/lbfgsb.f90:812-857,
:598-792) -- each run by the oracle's routine and fed, with the ORACLE's inputs, to the library's door
through import_state.  So every comparison is one routine deep: no difference is carried from one
routine into the next.

Bars: integers, iwhere, Index, Indx2, task strings, copies (t, r, the new W column) bit-exact; sums over
n rows 1e-12 of their scale (other summation order); results of the 2m x 2m solves 1e-9 relative to the
largest entry (conditioning, as in tests/test_gpu_parity.py).
"""
import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _fuzz_problem(seed, n, m):
    rng = np.random.default_rng(seed)
    a = 1.0 + 99.0 * rng.random(n)
    c = rng.normal(0, 2, n)
    wavy = seed % 2 == 1

    def fg(x, g):
        d = x - c
        f = 0.5 * np.sum(a * d * d)
        g[:] = a * d
        if wavy:
            f += np.sum(np.cos(3 * x))
            g[:] -= 3 * np.sin(3 * x)
        return float(f)
    l = rng.normal(-1, 1, n)
    u = l + np.abs(rng.normal(1.5, 1, n))
    fixed = rng.random(n) < 0.03
    u[fixed] = l[fixed]
    nbd = rng.integers(0, 4, n).astype(np.int32)
    x0 = rng.normal(0, 3, n)
    return po.Problem("routines%d" % seed, n, m, x0, l, u, nbd, 0.0, 0.0, fg, np.float64)


class Locals:
    """mainlb's saved locals out of lsave / isave / dsave (src/lbfgsb.f90:904-947)"""

    def __init__(self, s):
        i, d, ls = s.isave, s.dsave, s.lsave
        self.prjctd, self.cnstnd, self.boxed, self.updatd = (bool(v) for v in ls)
        self.head, self.col, self.itail, self.iter, self.iupdat = (int(i[k]) for k in (26, 27, 28, 29, 30))
        self.nfgv, self.nfree, self.ileave, self.nenter = int(i[33]), int(i[37]), int(i[39]), int(i[40])
        self.theta, self.epsmch, self.gd, self.sbgnrm = float(d[0]), float(d[4]), float(d[10]), float(d[12])
        self.stp, self.gdold, self.dtd = float(d[13]), float(d[14]), float(d[15])


def _parts(s):
    off = po.wa_offsets(s.n, s.m)
    w = {k: s.wa[o:o + sz].copy() for k, (o, sz) in off.items()}
    n = s.n
    w["index"], w["iwhere"], w["indx2"] = s.iwa[:n].copy(), s.iwa[n:2 * n].copy(), s.iwa[2 * n:].copy()
    return w


def _pack(s, w):
    """wa / iwa in the reference's layout from the parts"""
    off = po.wa_offsets(s.n, s.m)
    wa = np.zeros_like(s.wa)
    for k, (o, sz) in off.items():
        wa[o:o + sz] = w[k]
    iwa = np.concatenate([w["index"], w["iwhere"], w["indx2"]]).astype(np.int32)
    return wa, iwa


def _close(a, b, tol, what):
    a, b = np.asarray(a, float), np.asarray(b, float)
    scale = max(1.0, float(np.max(np.abs(b))) if b.size else 1.0)
    err = float(np.max(np.abs(a - b))) if b.size else 0.0
    assert err <= tol * scale, "%s: max |diff| %.3e over scale %.3e" % (what, err, scale)


def _chain_at(sol, torch, dev, R, p, s, checked):
    """one iteration's routines from the oracle state s (a NEW_X return), oracle and doors side by side"""
    n, m = p.n, p.m
    L = Locals(s)
    w = _parts(s)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    xd, gd_, ld, ud = T(s.x), T(s.g), T(p.l), T(p.u)
    nbdd = T(p.nbd.astype(np.int32))
    isave = s.isave.astype(np.int32)

    def imp(parts, nfree=None):
        wa, iwa = _pack(s, parts)
        isv = isave.copy()
        if nfree is not None:
            isv[37] = nfree
        sol.import_state(wa, iwa, isv)

    def exp():
        wa, iwa = sol.export_state()
        st = s.copy()
        st.wa, st.iwa = wa, iwa
        return _parts(st)

    # ------------------------------------------------------------------ mainlb :812-834 + matupd + formt
    stp = L.stp
    r_new = s.g - w["r"]
    rr = float(R.lib.lbo_ddot(n, po._ptr(r_new), po._ptr(r_new)))
    d_eff = w["d"].copy()
    if stp == 1.0:
        dr, ddum = L.gd - L.gdold, -L.gdold
    else:
        dr, ddum = (L.gd - L.gdold) * stp, -L.gdold * stp
        d_eff *= stp
    col, head, itail, iupdat, theta, updatd = L.col, L.head, L.itail, L.iupdat, L.theta, False
    o = dict(w)
    if dr > L.epsmch * ddum:
        updatd = True
        iupdat += 1
        ic, ih, it, th = (np.array([v], np.int32) for v in (col, head, itail, 0))
        th = np.array([theta])
        for k in ("ws", "wy", "sy", "ss", "wt"):
            o[k] = w[k].copy()
        R.matupd(n, m, o["ws"], o["wy"], o["sy"], o["ss"], d_eff, r_new, it, iupdat, ic, ih, th, rr, dr, stp,
                 L.dtd)
        col, head, itail, theta = int(ic[0]), int(ih[0]), int(it[0]), float(th[0])
        info = np.zeros(1, np.int32)
        R.formt(m, o["wt"], o["sy"], o["ss"], col, theta, info)
        assert info[0] == 0
        # the door, from the same inputs
        imp(w)
        ip = np.array([iupdat, L.col, L.head, L.itail], np.int32)
        th_g = sol.r_matupd(gd_, stp, dr, L.dtd, ip)
        e = exp()
        assert list(ip) == [iupdat, col, head, itail]
        assert abs(th_g - theta) <= 1e-12 * abs(theta), (th_g, theta)
        assert np.array_equal(e["ws"], o["ws"]) and np.array_equal(e["wy"], o["wy"]), "W columns"
        M = lambda a: a.reshape(m, m).T  # noqa: E731   (column-major m x m)
        _close(np.tril(M(e["sy"])[:col, :col]), np.tril(M(o["sy"])[:col, :col]), 1e-12, "sy")
        _close(np.triu(M(e["ss"])[:col, :col]), np.triu(M(o["ss"])[:col, :col]), 1e-12, "ss")
        checked["matupd"] += 1
    if col == 0:
        return
    # ------------------------------------------------------------------ cauchy
    oc = dict(o)
    oc["iwhere"] = w["iwhere"].copy()
    xcp = np.zeros(n)
    wk = [np.zeros(n, np.int32), np.zeros(n), np.zeros(n)]
    pc = [o["wa8m"][2 * m * k:2 * m * (k + 1)].copy() for k in range(4)]   # (work vectors: same stale content)
    nseg, info = np.zeros(1, np.int32), np.zeros(1, np.int32)
    R.cauchy(n, s.x, p.l, p.u, p.nbd, s.g, wk[0], oc["iwhere"], wk[1], wk[2], xcp, m, o["wy"], o["ws"], o["sy"],
             o["wt"], theta, col, head, pc[0], pc[1], pc[2], pc[3], nseg, L.sbgnrm, info, L.epsmch)
    assert info[0] == 0
    imp(o)
    xcp_d = torch.zeros(n, dtype=torch.float64, device=dev)
    ns_g, inf_g = sol.r_cauchy(xd, ld, ud, nbdd, gd_, theta, col, head, L.sbgnrm, xcp_d)
    e = exp()
    assert (ns_g, inf_g) == (int(nseg[0]), 0)
    assert np.array_equal(e["iwhere"], oc["iwhere"]), "iwhere after cauchy"
    _close(xcp_d.cpu().numpy(), xcp, 1e-12, "xcp")
    _close(e["z"], xcp, 1e-12, "z = xcp")
    for k, nm in enumerate(("p", "c", "wbp", "v")):
        _close(e["wa8m"][2 * m * k:2 * m * k + 2 * col], pc[k][:2 * col], 1e-9, "cauchy " + nm)
    checked["cauchy"] += 1
    oc["z"] = xcp
    oc["wa8m"] = np.concatenate(pc)
    # ------------------------------------------------------------------ freev
    of = dict(oc)
    of["index"], of["indx2"] = w["index"].copy(), w["indx2"].copy()
    nf, ne, il, wrk = (np.array([v], np.int32) for v in (L.nfree, 0, 0, 0))
    R.freev(n, nf, of["index"], ne, il, of["indx2"], oc["iwhere"], wrk, int(updatd), int(L.cnstnd), L.iter)
    nfree, nenter, ileave = int(nf[0]), int(ne[0]), int(il[0])
    imp(oc, nfree=L.nfree)
    got = sol.r_freev(L.iter, L.cnstnd, updatd)
    e = exp()
    if L.iter > 0 and L.cnstnd:
        assert got == (nfree, nenter, ileave, bool(wrk[0])), (got, nfree, nenter, ileave, wrk)
        assert np.array_equal(e["indx2"][:nenter], of["indx2"][:nenter]), "entering variables"
        assert np.array_equal(e["indx2"][ileave - 1:], of["indx2"][ileave - 1:]), "leaving variables"
    else:
        assert got[0] == nfree and got[3] == bool(wrk[0])
    assert np.array_equal(e["index"], of["index"]), "Index"
    checked["freev"] += 1
    if nfree == 0:
        return
    # ------------------------------------------------------------------ formk
    ok = dict(of)
    if wrk[0]:
        ok["wn"], ok["snd"] = of["wn"].copy(), of["snd"].copy()
        info[0] = 0
        R.formk(n, nfree, of["index"], nenter, ileave, of["indx2"], iupdat, int(updatd), ok["wn"], ok["snd"], m,
                o["ws"], o["wy"], o["sy"], theta, col, head, info)
        assert info[0] == 0
        imp(of, nfree=nfree)
        assert sol.r_formk(col, head, theta) == 0
        e = exp()
        M2 = lambda a: a.reshape(2 * m, 2 * m).T  # noqa: E731
        g1, o1 = M2(e["snd"]), M2(ok["snd"])
        for (r0, c0, lower) in ((0, 0, True), (m, m, True), (m, 0, False)):
            A, B = g1[r0:r0 + col, c0:c0 + col], o1[r0:r0 + col, c0:c0 + col]
            if lower:
                A, B = np.tril(A), np.tril(B)
            _close(A, B, 1e-11, "WN1 block (%d, %d)" % (r0, c0))
        _close(np.triu(M2(e["wn"])[:2 * col, :2 * col]), np.triu(M2(ok["wn"])[:2 * col, :2 * col]), 1e-8, "WN")
        checked["formk"] += 1
    # ------------------------------------------------------------------ cmprlb
    r = np.zeros(n)
    wa8 = ok["wa8m"].copy()
    info[0] = 0
    R.cmprlb(n, m, s.x, s.g, o["ws"], o["wy"], o["sy"], o["wt"], xcp, r, wa8, of["index"], theta, col, head,
             nfree, int(L.cnstnd), info)
    assert info[0] == 0
    imp(ok, nfree=nfree)
    r_d = torch.zeros(n, dtype=torch.float64, device=dev)
    assert sol.r_cmprlb(xd, gd_, theta, col, head, L.cnstnd, r_d) == 0
    r_full = np.zeros(n)
    rows = of["index"][:nfree] - 1
    r_full[rows] = r[:nfree]
    got_r = r_d.cpu().numpy()
    mask = np.zeros(n, bool)
    mask[rows] = True
    assert not got_r[~mask].any(), "r off the free rows"
    _close(got_r, r_full, 1e-11, "r of cmprlb")
    checked["cmprlb"] += 1
    # ------------------------------------------------------------------ subsm
    xs, ds, xps = xcp.copy(), r.copy(), np.zeros(n)
    iword = np.zeros(1, np.int32)
    wv = wa8[:2 * m].copy()
    info[0] = 0
    R.subsm(n, m, nfree, of["index"], p.l, p.u, p.nbd, xs, ds, xps, o["ws"], o["wy"], theta, s.x, s.g, col, head,
            iword, wv, ok["wn"], info)
    assert info[0] == 0
    imp(ok, nfree=nfree)
    xh_d = torch.zeros(n, dtype=torch.float64, device=dev)
    iw_g, inf_g = sol.r_subsm(xd, ld, ud, nbdd, gd_, T(r_full), theta, col, head, xh_d)
    assert (iw_g, inf_g) == (int(iword[0]), 0)
    _close(xh_d.cpu().numpy(), xs, 1e-9, "subspace minimiser")
    e = exp()
    _close(e["xp"], xcp, 1e-12, "xp = xcp")
    checked["subsm"] += 1
    if iword[0]:
        checked["subsm_bounded"] += 1
    # ------------------------------------------------------------------ lnsrlb: set-up call, then one more
    ol = dict(ok)
    ol["z"] = xs
    d_ls = xs - s.x                       # mainlb :720-722
    x_o, t_o, r_o = s.x.copy(), np.zeros(n), np.zeros(n)
    names = ("fold", "gd", "gdold", "stp", "dnorm", "dtd", "xstep", "stpmx")
    so = {k: np.zeros(1) for k in names}
    io = {k: np.zeros(1, np.int32) for k in ("ifun", "iback", "nfgv", "info")}
    io["nfgv"][0] = L.nfgv
    task_o, csave_o = po.pad60("NEW_X"), po.pad60("")
    is2_o, ds13_o = np.zeros(2, np.int32), np.zeros(13)
    f_now = float(s.f[0])

    def oracle_ls(f, g):
        R.lnsrlb(n, p.l, p.u, p.nbd, x_o, f, so["fold"], so["gd"], so["gdold"], g, d_ls, r_o, t_o, xs, so["stp"],
                 so["dnorm"], so["dtd"], so["xstep"], so["stpmx"], L.iter, io["ifun"], io["iback"], io["nfgv"],
                 io["info"], task_o, int(L.boxed), int(L.cnstnd), csave_o, is2_o, ds13_o)
    imp(ol, nfree=nfree)
    x_g = xd.clone()
    sc = np.zeros(8)
    ic = np.array([L.iter, 0, 0, L.nfgv, 0, int(L.boxed), int(L.cnstnd)], np.int32)
    task_g, csave_g = po.pad60("NEW_X"), po.pad60("")
    is2_g, ds13_g = np.zeros(2, np.int32), np.zeros(13)

    def compare_ls(tag):
        for k, nm in enumerate(names):
            a, b = sc[k], float(so[nm][0])
            assert abs(a - b) <= 1e-12 * max(1.0, abs(b)), "%s %s: %r vs %r" % (tag, nm, a, b)
        assert [int(v) for v in ic[1:5]] == [int(io[k][0]) for k in ("ifun", "iback", "nfgv", "info")], tag
        assert po.task_str(task_g) == po.task_str(task_o), tag
        assert po.task_str(csave_g) == po.task_str(csave_o), tag
        assert list(is2_g) == list(is2_o), tag
        _close(ds13_g, ds13_o, 1e-12, tag + " dsave of dcsrch")
        _close(x_g.cpu().numpy(), x_o, 1e-13, tag + " trial point")
    oracle_ls(f_now, s.g)
    sol.r_lnsrlb(x_g, ld, ud, nbdd, gd_, f_now, sc, ic, task_g, csave_g, is2_g, ds13_g)
    compare_ls("set-up call")
    e = exp()
    assert np.array_equal(e["t"], s.x) and np.array_equal(e["r"], s.g), "t = x, r = g"
    assert np.array_equal(e["d"], d_ls), "d = z - x"
    checked["lnsrlb"] += 1
    if po.task_str(task_o).startswith("FG_LN"):
        g2 = np.zeros(n)
        f2 = p.fg(x_o.copy(), g2)
        oracle_ls(f2, g2)
        sol.r_lnsrlb(x_g, ld, ud, nbdd, T(g2), f2, sc, ic, task_g, csave_g, is2_g, ds13_g)
        compare_ls("second call")
        checked["lnsrlb_second"] += 1


CASES = [
    # seed, n, m, iterations to run, states examined
    (101, 3000, 5, 14),
    (102, 5000, 17, 24),
    (103, 2500, 40, 46),     # more pairs than the fused kernels hold: every W product in tiles
    (104, 4000, 10, 16),
    (105, 150_000, 10, 9),   # many workgroups per reduction, thousands of breakpoints per walk
]


@pytest.mark.parametrize("seed,n,m,iters", CASES)
def test_routine_doors_along_a_trajectory(oracle_built, seed, n, m, iters):
    import torch
    import lbfgsb_amd
    dev = torch.device("cuda", 0)
    p = _fuzz_problem(seed, n, m)
    eng = po.Engine("oracle")
    states = []
    po.run(eng, p, max_iter=iters,
           snapshot=lambda k, s: states.append(s.copy()) if s.task_s.startswith("NEW_X") else None)
    assert len(states) >= min(iters, 6)
    R = po.Routines()
    checked = {k: 0 for k in ("matupd", "cauchy", "freev", "formk", "cmprlb", "subsm", "subsm_bounded", "lnsrlb",
                              "lnsrlb_second")}
    sol = lbfgsb_amd.DeviceSolver(n, m, device=0, mirror_index=True)
    try:
        pick = sorted(set([0, 1, 2, 3, len(states) // 2, len(states) - 2, len(states) - 1]) & set(range(len(states))))
        for k in pick:
            _chain_at(sol, torch, dev, R, p, states[k], checked)
    finally:
        sol.close()
    print("routine doors checked:", checked)
    for k in ("matupd", "cauchy", "freev", "formk", "cmprlb", "subsm", "lnsrlb"):
        assert checked[k] >= 3, checked


def test_active_and_errclb_doors(oracle_built):
    import torch
    import lbfgsb_amd
    dev = torch.device("cuda", 0)
    R = po.Routines()
    p = _fuzz_problem(7, 4000, 5)
    n = p.n
    sol = lbfgsb_amd.DeviceSolver(n, p.m, device=0, mirror_index=True)
    try:
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        x_o, iw_o = p.x0.copy(), np.zeros(n, np.int32)
        fl = [np.zeros(1, np.int32) for _ in range(4)]
        R.active(n, p.l, p.u, p.nbd, x_o, iw_o, *fl)
        x_g = T(p.x0)
        got = sol.r_active(x_g, T(p.l), T(p.u), T(p.nbd))
        assert got == tuple(bool(v[0]) for v in fl[:3])
        assert np.array_equal(x_g.cpu().numpy(), x_o)
        _, iwa = sol.export_state()
        assert np.array_equal(iwa[n:2 * n], iw_o)
        # errclb: valid input, an invalid nbd, an infeasible box, both (the reference reports the last offender)
        for bad_nbd, bad_box in ((None, None), (1234, None), (None, 77), (100, 3000), (3000, 100)):
            l, u, nbd = p.l.copy(), p.u.copy(), p.nbd.copy()
            if bad_nbd is not None:
                nbd[bad_nbd] = 5
            if bad_box is not None:
                nbd[bad_box], l[bad_box], u[bad_box] = 2, 1.0, 0.0
            task_o = po.pad60("START")
            info, k = np.zeros(1, np.int32), np.zeros(1, np.int32)
            R.errclb(n, p.m, 1e7, l, u, nbd, task_o, info, k)
            t_g, info_g, k_g = sol.r_errclb(T(l), T(u), T(nbd), 1e7)
            assert t_g == po.task_str(task_o)
            assert (info_g, k_g) == (int(info[0]), int(k[0]) if info[0] else 0)
    finally:
        sol.close()


def test_cauchy_door_reference_nan_point(oracle_built):
    """The reference's own arithmetic on a degenerate state: x an ulp OUTSIDE the box (what a line-search
    step of stpmx can leave behind), so that the projected gradient is a few ulps while no variable can
    move -- d == 0, f1 = f2 = 0, dtm = 0/0, and xcp = x + tsum * d is NaN in EVERY component
    (src/lbfgsb.f90:1357-1365, :1509-1515).  The door must return the same thing, not a tidier one; and a
    variable found beyond its bound by the scan keeps its x in xcp, not the bound (:1284-1291)."""
    import torch
    import lbfgsb_amd
    dev = torch.device("cuda", 0)
    R = po.Routines()
    n, m = 300, 4
    rng = np.random.default_rng(5)
    nbd = rng.integers(1, 4, n).astype(np.int32)
    l, u = -np.ones(n), np.ones(n)
    x = np.where(nbd == 3, u, l).astype(float)          # everybody on a bound ...
    g = np.where(nbd == 3, -1.0, 1.0) * (0.5 + rng.random(n))   # ... pushed outwards
    nbd[:40] = 0                                         # free variables with zero gradient
    x[:40], g[:40] = rng.normal(0, 1, 40), 0.0
    k = 77
    nbd[k], x[k], g[k] = 1, np.nextafter(l[k], -np.inf), 0.3   # an ulp below its lower bound
    for kind in ("nan", "finite"):
        if kind == "finite":
            g[5] = -0.25                                  # one variable moves: a finite Cauchy point
        sbg = float(R.projgr(n, l, u, nbd, x, g))
        assert sbg > 0.0
        iw_o = np.where(nbd == 0, -1, 0).astype(np.int32)   # iwhere as active (:1024-1037) leaves it
        xcp = np.zeros(n)
        wk = [np.zeros(n, np.int32), np.zeros(n), np.zeros(n)]
        pc = [np.zeros(2 * m) for _ in range(4)]
        ws = np.zeros(m * n)
        sy = np.zeros(m * m)
        nseg, info = np.zeros(1, np.int32), np.zeros(1, np.int32)
        iw_ref = iw_o.copy()
        R.cauchy(n, x, l, u, nbd, g, wk[0], iw_ref, wk[1], wk[2], xcp, m, ws, ws, sy, sy, 1.0, 0, 1, pc[0], pc[1],
                 pc[2], pc[3], nseg, sbg, info, float(np.finfo(float).eps))
        assert np.isnan(xcp).all() if kind == "nan" else np.isfinite(xcp).all()
        sol = lbfgsb_amd.DeviceSolver(n, m, device=0, mirror_index=True)
        try:
            T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
            sol.set_iwhere(iw_o)
            xcp_d = torch.zeros(n, dtype=torch.float64, device=dev)
            ns_g, inf_g = sol.r_cauchy(T(x), T(l), T(u), T(nbd), T(g), 1.0, 0, 1, sbg, xcp_d)
            got = xcp_d.cpu().numpy()
            _, iwa = sol.export_state()
        finally:
            sol.close()
        assert (ns_g, inf_g) == (int(nseg[0]), int(info[0]))
        assert np.array_equal(iwa[n:2 * n], iw_ref)
        if kind == "nan":
            assert np.isnan(got).all(), "the reference's Cauchy point is NaN in every component here"
        else:
            assert np.array_equal(got, xcp), "finite case: bit for bit (no sums over rows with col = 0)"
            assert got[k] == x[k] < l[k]      # found beyond its bound: x stays where it is


@pytest.mark.parametrize("real", [np.float64, np.float32])
def test_level1_doors(oracle_built, real):
    """vec_sub / vec_scale / dot (src/lbfgsb_blas_module.F90:37-277 at the n-length call sites
    of src/lbfgsb.f90:720-722, :812-822).  Differences and scalings are one rounding each: bit-exact against
    numpy in the context's real kind, in place too.  The dot against the oracle's ddot (sequential, groups of
    five): other order of summation, 1e-12 (fp64 accumulation; REAL32: the reference accumulates in REAL32) of
    sum |a_i b_i|."""
    import torch
    import lbfgsb_amd
    twin = po.Routines(real)     # (sets lbo_ddot's prototype)
    tdt = torch.float32 if real == np.float32 else torch.float64
    for seed, n in enumerate((1, 2, 3, 5, 63, 64, 255, 1000, 4097, 1_000_003)):
        rng = np.random.default_rng(100 + seed)
        a = (rng.normal(0, 1, n) * 10.0 ** rng.uniform(-6, 6, n)).astype(real)
        b = (rng.normal(0, 1, n) * 10.0 ** rng.uniform(-6, 6, n)).astype(real)
        sol = lbfgsb_amd.DeviceSolver(n, 3, device=0, real32=real == np.float32)
        try:
            da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
            out = torch.full((n,), 9.0, dtype=tdt, device="cuda")
            torch.cuda.synchronize()             # (the doors run on the context's stream, torch on its own)
            sol.r_vec_sub(da, db, out)
            assert np.array_equal(out.cpu().numpy(), a - b)
            out.copy_(da)
            torch.cuda.synchronize()
            sol.r_vec_sub(out, db, out)          # in place
            assert np.array_equal(out.cpu().numpy(), a - b)
            alpha = 0.2992887057956888
            out.copy_(db)
            torch.cuda.synchronize()
            sol.r_vec_scale(alpha, out)
            assert np.array_equal(out.cpu().numpy(), real(alpha) * b)
            got = sol.r_dot(da, db)
            ref = float(twin.lib.lbo_ddot(n, a.ctypes.data, b.ctypes.data))
            exact = float(np.sum(a.astype(np.longdouble) * b.astype(np.longdouble)))
            scale = float(np.sum(np.abs(a.astype(np.float64) * b.astype(np.float64))))
            assert abs(got - exact) <= 1e-12 * scale, (n, got, exact)
            assert abs(got - ref) <= (1e-12 if real == np.float64 else 4e-7 * max(1.0, np.sqrt(n))) * scale, (n, got, ref)
            assert sol.r_dot(da, da) >= 0.0
        finally:
            sol.close()

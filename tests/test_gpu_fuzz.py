"""Randomised differential test: random box-constrained problems (random n, m, bound types incl.
fixed and unbounded variables, separable + coupled + non-convex objectives, random factr/pgtol)
solved call by call through the reference-shaped host entry and by the oracle.  Every setulb
return must match (task, iteration, nfg, nseg, nfree; f to 1e-8) -- except that a run may part
ways late (second half), where with factr = 0 the stop test acts on rounding noise, provided
the final f agrees to 1e-7.  Exercises the production iteration (speculative update pass, pending
pair, functional Cauchy point, skipped updates, restarts) on shapes no hand-written case covers."""
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


@pytest.mark.parametrize("first,count,nmax,mlo,mhi,switch", [
    (0, 120, 400, 1, 13, None), (5000, 40, 3000, 11, 33, None),
    # the measurement switches select fallback paths that must stay correct: the candidate
    # hand-over of the update pass, and the three-pass iteration (no closed form, stored z and d)
    (7000, 40, 1500, 1, 25, "LBFGSB_SPEC_CAPTURE=1"), (7100, 40, 1500, 1, 25, "LBFGSB_TWO_PASS=0"),
    (7200, 40, 1500, 1, 25, "LBFGSB_LEAN=0")])
def test_random_problems_against_oracle(oracle_built, monkeypatch, first, count, nmax, mlo, mhi, switch):
    po = oracle_built
    import lbfgsb_amd as la
    if switch:
        monkeypatch.setenv(*switch.split("="))

    def row(s):
        return (s.task_s[:12], int(s.isave[29]), int(s.isave[33]), int(s.isave[32]), int(s.isave[37]),
                float(s.f[0]))
    late = 0
    for seed in range(first, first + count):
        p = make(po, seed, nmax, mlo, mhi)
        ro = []
        so = po.run(po.Engine("oracle"), p, max_iter=80, snapshot=lambda k, s: ro.append(row(s)))
        s = po.State.fresh(p)
        nbd = p.nbd.astype(np.int32)
        rg = []
        for _ in range(100000):
            la.setulb(p.n, p.m, s.x, p.l, p.u, nbd, s.f, s.g, p.factr, p.pgtol, s.wa, s.iwa, s.task,
                      -1, s.csave, s.lsave, s.isave, s.dsave)
            rg.append(row(s))
            t = s.task_s
            if t.startswith("FG"):
                s.f[0] = p.fg(s.x, s.g)
            elif t.startswith("NEW_X"):
                if s.isave[29] >= 80:
                    break
            else:
                break
        k = 0
        while (k < min(len(ro), len(rg)) and ro[k][:5] == rg[k][:5]
               and abs(ro[k][5] - rg[k][5]) <= 1e-8 * max(1.0, abs(ro[k][5]))):
            k += 1
        if k == len(ro) == len(rg):
            continue
        late += 1
        fo, fgp = float(so.f[0]), float(s.f[0])
        assert k >= 0.4 * len(ro) and abs(fo - fgp) <= 1e-7 * max(1.0, abs(fo)), \
            (seed, p.n, p.m, k, len(ro), len(rg), ro[max(0, k - 1):k + 1], rg[max(0, k - 1):k + 1])
    assert late <= 0.2 * count      # the vast majority match through the last call


def test_random_problems_parallel_gcp_search(oracle_built, monkeypatch):
    """The opt-in parallel GCP search on random problems: LBFGSB_PG_MIN=0 sends EVERY walk that
    passes its first breakpoint through it -- the closed form when no pair is stored, the sort +
    scans (with the f2 clamp) when pairs are stored -- on all bound types, fixed and unbounded
    variables, m = 1..12.  Against the oracle's sequential walk: same iteration / nfg columns,
    nseg and nfree within 2, f to 1e-8, call by call over the first 12
    iterations; a run may part ways late only at rounding level (final f to 1e-7)."""
    po = oracle_built
    import torch
    import lbfgsb_amd as la
    monkeypatch.setenv("LBFGSB_PG_MIN", "0")
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
        sol = la.DeviceSolver(p.n, p.m, parallel_gcp=True)
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


def test_random_problems_exact_tie_order(oracle_built, monkeypatch):
    """LBFGSB_F_EXACT_TIES with LBFGSB_EXACT_ALWAYS=1: EVERY walk of 40 random problems is replayed
    in the order of the reference's heap (all breakpoint times on the host, hpsolb, records
    gathered in pop order) instead of the device's (t, index) order.  Same bar as the main
    differential test: every NEW_X row (iteration, nfg, nseg, nfree; f to 1e-8) equals the
    oracle's; a run may part ways late only at rounding level."""
    po = oracle_built
    import torch
    import lbfgsb_amd as la
    monkeypatch.setenv("LBFGSB_EXACT_ALWAYS", "1")
    late, count = 0, 40
    for seed in range(9500, 9500 + count):
        p = make(po, seed, 1500, 1, 13)
        ro = []
        so = po.run(po.Engine("oracle"), p, max_iter=30,
                    snapshot=lambda k, s: ro.append((int(s.isave[29]), int(s.isave[33]), int(s.isave[32]),
                                                     int(s.isave[37]), float(s.f[0])))
                    if s.task_s.startswith("NEW_X") else None)
        sol = la.DeviceSolver(p.n, p.m, exact_ties=True)
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
                if sol.isave[29] >= 30:
                    break
            else:
                break
        fgp = float(sol.f[0])
        sol.close()
        k = 0
        while (k < min(len(ro), len(rg)) and ro[k][:4] == rg[k][:4]
               and abs(ro[k][4] - rg[k][4]) <= 1e-8 * max(1.0, abs(ro[k][4]))):
            k += 1
        if k == len(ro) == len(rg):
            continue
        late += 1
        fo = float(so.f[0])
        assert k >= 0.4 * len(ro) and abs(fo - fgp) <= 1e-7 * max(1.0, abs(fo)), \
            (seed, p.n, p.m, k, len(ro), len(rg), ro[max(0, k - 1):k + 1], rg[max(0, k - 1):k + 1])
    assert late <= 0.2 * count

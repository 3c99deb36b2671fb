"""Boundary behaviour of the C ABI that the parity tests do not touch: stream ordering of the
caller's f,g evaluation, the registry behind the host-pointer form (the reference's own
argument list, src/lbfgsb.f90:88-89), error propagation out of the minimize wrapper."""
import ctypes as C

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


def test_fg_produced_on_another_stream_is_ordered(env):
    """include/lbfgsb_hip.h "Stream ordering": g written by work that is still QUEUED on another
    stream when setulb is re-entered (no host synchronisation in between) must be what the
    solver reads -- DeviceSolver.setulb orders the solver's stream behind torch's current stream
    (lbfgsb_hip_wait_stream).  Same trajectory as the fully synchronised loop."""
    po, torch, la = env["po"], env["torch"], env["la"]
    n, m, iters = 200_003, 5, 6
    p = po.problem_quadratic(n, m, mixed_nbd=True)
    i = torch.arange(1, n + 1, dtype=torch.int64, device="cuda")
    a = 1.0 + 99.0 * ((7919 * i) % 10007).double() / 10006.0
    c = -2.0 + 4.0 * ((104729 * i) % 100003).double() / 100002.0

    def run(async_side):
        sol = la.DeviceSolver(n, m)
        x = torch.from_numpy(p.x0.copy()).cuda()
        g = torch.zeros_like(x)
        l, u = torch.from_numpy(p.l).cuda(), torch.from_numpy(p.u).cuda()
        nbd = torch.from_numpy(p.nbd.astype(np.int32)).cuda()
        side = torch.cuda.Stream()
        rows = []
        while True:
            if async_side and sol.task_s.startswith("FG"):
                with torch.cuda.stream(side):
                    t = sol.setulb(x, l, u, nbd, g, 0.0, 0.0)
            else:
                t = sol.setulb(x, l, u, nbd, g, 0.0, 0.0)
            if t.startswith("FG"):
                if async_side:
                    # the FG return synchronised the solver's stream: x is complete.  f is taken
                    # synchronously; g is written on a side stream behind a long spin, so that it
                    # is still being produced when setulb is called again
                    d = x - c
                    sol.f[0] = float(0.5 * torch.sum(a * d * d))
                    with torch.cuda.stream(side):
                        torch.cuda._sleep(20_000_000)
                        g.copy_(a * (x - c))
                else:
                    d = x - c
                    g.copy_(a * d)
                    sol.f[0] = float(0.5 * torch.sum(a * d * d))
                    torch.cuda.synchronize()
            elif t.startswith("NEW_X"):
                rows.append((int(sol.isave[29]), int(sol.isave[33]), int(sol.isave[32]), int(sol.isave[37]),
                             float(sol.f[0])))
                if sol.isave[29] >= iters:
                    break
            else:
                break
        torch.cuda.synchronize()
        sol.close()
        return rows
    ref = run(False)
    got = run(True)
    assert [r[:4] for r in got] == [r[:4] for r in ref]
    for a_, b_ in zip(got, ref):
        assert a_[4] == pytest.approx(b_[4], rel=1e-12)


def test_host_form_registry_and_offsets(env):
    """The host-pointer form keeps its context in a registry keyed by isave(17:18): stale or
    garbage handles are refused, a caller that stops by itself releases the context with
    lbfgsb_hip_release_host, START over a live run frees the old one, and isave(1:16) carry the
    wa offsets the reference persists there (src/lbfgsb.f90:250-265)."""
    po, la = env["po"], env["la"]
    lib = la.load_library()
    p = po.problem_quadratic(300, 4, mixed_nbd=True)
    nbd = p.nbd.astype(np.int32)

    def call(s):
        la.setulb(p.n, p.m, s.x, p.l, p.u, nbd, s.f, s.g, 0.0, 0.0, s.wa, s.iwa, s.task, -1, s.csave,
                  s.lsave, s.isave, s.dsave)
        if s.task_s.startswith("FG"):
            s.f[0] = p.fg(s.x, s.g)
    s = po.State.fresh(p)
    for _ in range(6):
        call(s)
    n, m = p.n, p.m
    mn, mm = m * n, m * m
    want = [mn, mm, 4 * mm, 1, 1 + mn, 1 + 2 * mn, 1 + 2 * mn + mm, 1 + 2 * mn + 2 * mm, 1 + 2 * mn + 3 * mm,
            1 + 2 * mn + 7 * mm, 1 + 2 * mn + 11 * mm]
    assert s.isave[:11].tolist() == want
    assert s.isave[11] == want[10] + n and s.isave[15] == want[10] + 5 * n       # lr ... lwa
    # the caller stops by itself (driver2 style) and releases the context
    assert lib.lbfgsb_hip_release_host(s.isave.ctypes.data_as(C.c_void_p)) == 0
    assert s.isave[16] == 0 and s.isave[17] == 0
    with pytest.raises(la.LbfgsbError):
        call(s)                                   # the run is gone: refused, not dereferenced
    # garbage in the handle slots
    s2 = po.State.fresh(p)
    call(s2)
    bad = s2.copy()
    bad.isave[16] = 123456
    with pytest.raises(la.LbfgsbError):
        call(bad)
    bad = s2.copy()
    bad.isave[17] = 77
    with pytest.raises(la.LbfgsbError):
        call(bad)
    # START over the isave of a live run: the old context is freed, the new run is independent
    old_id = int(s2.isave[16])
    s2.task[:] = po.pad60("START")
    s2.x[:] = p.x0
    call(s2)
    assert s2.task_s.startswith("FG_START") and int(s2.isave[16]) != 0
    stale = s2.copy()
    stale.isave[16] = old_id if old_id != int(s2.isave[16]) else old_id + 1000
    with pytest.raises(la.LbfgsbError):
        call(stale)
    lib.lbfgsb_hip_release_host(s2.isave.ctypes.data_as(C.c_void_p))


def test_minimize_callback_exception_is_raised(env):
    """A Python exception inside the fg callback of DeviceSolver.minimize must not be swallowed
    by the C frame: the loop ends and the exception is raised to the caller."""
    po, torch, la = env["po"], env["torch"], env["la"]
    n, m = 1001, 4
    p = po.problem_quadratic(n, m)
    sol = la.DeviceSolver(n, m)
    x = torch.from_numpy(p.x0.copy()).cuda()
    g = torch.zeros_like(x)
    l, u = torch.from_numpy(p.l).cuda(), torch.from_numpy(p.u).cuda()
    nbd = torch.from_numpy(p.nbd.astype(np.int32)).cuda()
    calls = []

    def fg(xp, gp):
        calls.append(1)
        if len(calls) == 3:
            raise ValueError("objective blew up")
        return sol.objective(0, x, g)
    with pytest.raises(ValueError, match="objective blew up"):
        sol.minimize(x, l, u, nbd, g, fg=fg, factr=0.0, pgtol=0.0, max_iter=50)
    assert sol.task_s.startswith("STOP: THE OBJECTIVE CALLBACK RETURNED NaN")
    assert len(calls) == 3
    sol.close()


def test_ping_pong_entry_is_bit_identical_to_the_classic_one():
    """lbfgsb_hip_setulb_dev_pp against lbfgsb_hip_setulb_dev on the same problems: the two entries
    differ only in WHERE t, r and the trial point live (roles of two caller buffer pairs instead of
    copies), so every return must be identical bit for bit -- task, every isave / dsave slot, f -- and so
    must the reference-layout state (export_state: z, r, d, t, Ws, Wy, the m x m matrices, iwhere) and
    the iterate itself.  Problems: the bounded quadratic with all four bound types, box-bounded
    Rosenbrock (rejected trials, non-unit steps), an unconstrained and an all-but-fixed problem, and
    random problems that take restarts and the backtracking branch of subsm."""
    import numpy as np
    import torch
    import lbfgsb_amd as la
    from oracle import pyoracle as po
    from test_gpu_fuzz import make
    TIME_D = [5, 6, 7, 8, 9]
    problems = [po.problem_quadratic(4099, 7, mixed_nbd=True), po.problem_rosenbrock(1000, 10, 0.0, 0.0),
                po.problem_rosenbrock(25, 5, 1e7, 1e-5)]
    q = po.problem_quadratic(777, 5)
    q.nbd[:] = 0
    problems.append(q)
    problems += [make(po, seed, 600, 1, 25) for seed in range(11000, 11040)]
    swaps = 0
    for p in problems:
        def run(pp):
            sol = la.DeviceSolver(p.n, p.m)
            xs = [torch.from_numpy(p.x0.copy()).cuda(), torch.full((p.n,), 5.0, dtype=torch.float64, device="cuda")]
            gs = [torch.zeros_like(xs[0]), torch.full_like(xs[0], 9.0)]
            l, u = torch.from_numpy(p.l).cuda(), torch.from_numpy(p.u).cuda()
            nbd = torch.from_numpy(p.nbd.astype(np.int32)).cuda()
            x, g, cur, trace, curs = xs[0], gs[0], 0, [], []
            for _ in range(400):
                if pp:
                    t, cur = sol.setulb_pp(xs, l, u, nbd, gs, p.factr, p.pgtol)
                    x, g = xs[cur], gs[cur]
                else:
                    t = sol.setulb(x, l, u, nbd, g, p.factr, p.pgtol)
                torch.cuda.synchronize()
                wa, iwa = sol.export_state()
                ds = sol.dsave.copy()
                ds[TIME_D] = 0
                trace.append((t, sol.isave[21:44].copy(), ds, float(sol.f[0]), wa, iwa, x.cpu().numpy(),
                              g.cpu().numpy()))
                curs.append(cur)
                if t.startswith("FG"):
                    xh = x.cpu().numpy()
                    gh = np.empty_like(xh)
                    sol.f[0] = p.fg(xh, gh)
                    g.copy_(torch.from_numpy(gh))
                elif t.startswith("NEW_X"):
                    if sol.isave[29] >= 40:
                        break
                else:
                    break
            sol.close()
            return trace, curs
        a, _ = run(False)
        b, curs = run(True)
        swaps += sum(1 for c0, c1 in zip(curs, curs[1:]) if c0 != c1)
        assert len(a) == len(b), (p.name, len(a), len(b))
        for k, (ra, rb) in enumerate(zip(a, b)):
            assert ra[0] == rb[0], (p.name, k, ra[0], rb[0])
            assert np.array_equal(ra[1], rb[1]), (p.name, k, ra[1], rb[1])
            assert ra[2].tobytes() == rb[2].tobytes() and ra[3] == rb[3], (p.name, k)
            assert ra[4].tobytes() == rb[4].tobytes(), (p.name, k, "wa differs", ra[0])
            assert np.array_equal(ra[5], rb[5]), (p.name, k, "iwa differs")
            assert ra[6].tobytes() == rb[6].tobytes(), (p.name, k, "x")
            # (at an 'FG' return g[cur] is the buffer the caller is about to write the gradient into)
            assert ra[0].startswith("FG") or ra[7].tobytes() == rb[7].tobytes(), (p.name, k, "g")
    assert swaps > 500      # the pairs really did change roles


def test_m_beyond_the_limit_is_answered_like_an_argument_error():
    """The reference puts no upper limit on m (src/lbfgsb.f90:93-97); this library takes m <= LBFGSB_MAX_M
    = 1024 (fused kernels up to 32 pairs, unfused tiles beyond: test_gpu_parity::test_wide_memory_*).
    Through the reference-shaped host entry a larger m is answered the way the reference answers its own
    argument errors -- task = 'ERROR: ...', no iteration, return code 0 -- in both integer widths; the
    context entry refuses it with LBFGSB_E_ARG."""
    import ctypes as C
    import numpy as np
    import lbfgsb_amd as la
    from lbfgsb_amd import capi
    from oracle import pyoracle as po
    n, m = 50, 1025
    p = po.problem_quadratic(n, m)
    s = po.State.fresh(p)
    la.setulb(n, m, s.x, p.l, p.u, p.nbd.astype(np.int32), s.f, s.g, 0.0, 0.0, s.wa, s.iwa, s.task, -1,
              s.csave, s.lsave, s.isave, s.dsave)
    assert s.task_s == "ERROR: M > 1024 (LIMIT OF LBFGSB_HIP)"
    lib = la.load_library()
    s8 = po.State.fresh(p, np.int64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    rc = lib.lbfgsb_hip_setulb_host_ik(n, m, vp(s8.x), vp(p.l), vp(p.u), vp(p.nbd.astype(np.int64)), vp(s8.f),
                                       vp(s8.g), 0.0, 0.0, vp(s8.wa), vp(s8.iwa), vp(s8.task), -1, vp(s8.csave),
                                       vp(s8.lsave), vp(s8.isave), vp(s8.dsave), None, 8, 0, 8)
    assert rc == 0 and s8.task_s == "ERROR: M > 1024 (LIMIT OF LBFGSB_HIP)"
    h = C.c_void_p()
    assert lib.lbfgsb_hip_create(n, n, 0, m, 0, 0, None, C.byref(h)) == -101   # LBFGSB_E_ARG


def test_host_entry_with_eight_byte_integers_matches_the_four_byte_one():
    """lbfgsb_hip_setulb_host_ik with int_bytes = 8 (what a -fdefault-integer-8 Fortran build binds)
    against the int32 entry on the same problem: same task sequence, same counters in isave(22:44), same
    x, f to the last bit; isave(1:16) carry the wa offsets in full width."""
    import ctypes as C
    import numpy as np
    import lbfgsb_amd as la
    from oracle import pyoracle as po
    lib = la.load_library()
    p = po.problem_quadratic(3001, 6, mixed_nbd=True)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731

    def run(ib):
        it = np.int64 if ib == 8 else np.int32
        s = po.State.fresh(p, it)
        nbd = p.nbd.astype(it)
        rows = []
        for _ in range(500):
            rc = lib.lbfgsb_hip_setulb_host_ik(p.n, p.m, vp(s.x), vp(p.l), vp(p.u), vp(nbd), vp(s.f), vp(s.g),
                                               0.0, 0.0, vp(s.wa), vp(s.iwa), vp(s.task), -1, vp(s.csave),
                                               vp(s.lsave), vp(s.isave), vp(s.dsave), None, 8, 0, ib)
            assert rc == 0
            t = s.task_s
            rows.append((t, [int(v) for v in s.isave[21:44]], [int(v) for v in s.lsave], float(s.f[0]), s.x.copy()))
            if t.startswith("FG"):
                s.f[0] = p.fg(s.x, s.g)
            elif t.startswith("NEW_X"):
                if s.isave[29] >= 12:
                    s.task[:] = po.pad60("STOP: done")
            else:
                break
        return rows, s
    r4, s4 = run(4)
    r8, s8 = run(8)
    assert len(r4) == len(r8) and r4[-1][0].startswith("STOP")
    for a, b in zip(r4, r8):
        assert a[:4] == b[:4] and a[4].tobytes() == b[4].tobytes()
    mn, mm = p.m * p.n, p.m * p.m
    assert list(s8.isave[:4]) == [mn, mm, 4 * mm, 1] and s8.isave[15] == 1 + 2 * mn + 11 * mm + 5 * p.n
    assert list(s4.isave[:16]) == list(s8.isave[:16])      # (no saturation at this size)


def test_uniform_bounds_are_detected_and_change_nothing():
    """Bound arrays that hold one value each are read as constants by the passes over W
    (lbfgsb_hip_uniform_bounds): detection per array (l, u, nbd independently); arrays with FEW distinct values
    (<= 8 each: driver3's alternating box, test/driver3.f90:102-120) are dictionary-coded (bit 3: the packed nbd
    byte carries the table indices, mask = 1 | 2 | 8); a 9th value falls back to streaming.  The trajectory --
    every isave / dsave slot, f, x, the exported state -- is bit for bit the one of a context with the
    detection switched off."""
    import numpy as np
    import torch
    import lbfgsb_amd as la
    from oracle import pyoracle as po
    TIME_D = [5, 6, 7, 8, 9]
    cases = []
    cases.append((po.problem_quadratic(5003, 7), 7))                      # l, u, nbd all uniform
    cases.append((po.problem_quadratic(4099, 6, mixed_nbd=True), 3))      # nbd varies
    cases.append((po.problem_rosenbrock(1000, 10, 0.0, 0.0), 11))         # l alternates (2 values): dictionary
    q = po.problem_quadratic(3001, 12)
    q.u[17] = np.nextafter(1.0, 2.0)                                      # one entry differs in the last bit
    cases.append((q, 11))                                                 # ... a 2-entry table for u
    q = po.problem_quadratic(2000, 5)
    q.l[:] = -0.0
    q.l[3] = 0.0                                                          # -0.0 vs 0.0 count as different
    q.u[:] = 2.0
    cases.append((q, 11))
    rng = np.random.default_rng(5)
    for nl, nu, mixed, want in ((3, 1, False, 11), (8, 8, True, 11), (9, 1, False, 6), (2, 9, True, 0),
                                (9, 9, False, 4)):
        q = po.problem_quadratic(6007, 8, mixed_nbd=mixed)
        lv = -1.0 - 0.125 * np.arange(nl)                                 # nl distinct lower, nu distinct upper bounds
        uv = 1.0 + 0.25 * np.arange(nu)
        q.l[:] = lv[rng.integers(0, nl, q.n)]
        q.u[:] = uv[rng.integers(0, nu, q.n)]
        q.l[:nl], q.u[:nu] = lv, uv                                       # (every value occurs)
        cases.append((q, want))
    for p, want_mask in cases:
        def run(on):
            sol = la.DeviceSolver(p.n, p.m, options={"uniform_bounds": 1 if on else 0})
            xs = [torch.from_numpy(p.x0.copy()).cuda(), torch.zeros(p.n, dtype=torch.float64, device="cuda")]
            gs = [torch.zeros_like(xs[0]), torch.zeros_like(xs[0])]
            l, u = torch.from_numpy(p.l).cuda(), torch.from_numpy(p.u).cuda()
            nbd = torch.from_numpy(p.nbd.astype(np.int32)).cuda()
            trace, mask = [], None
            for _ in range(300):
                t, cur = sol.setulb_pp(xs, l, u, nbd, gs, p.factr, p.pgtol)
                if mask is None:
                    mask = sol.uniform_bounds()
                torch.cuda.synchronize()
                wa, iwa = sol.export_state()
                ds = sol.dsave.copy()
                ds[TIME_D] = 0
                trace.append((t, sol.isave[21:44].copy(), ds, float(sol.f[0]), wa, iwa, xs[cur].cpu().numpy()))
                if t.startswith("FG"):
                    xh = xs[cur].cpu().numpy()
                    gh = np.empty_like(xh)
                    sol.f[0] = p.fg(xh, gh)
                    gs[cur].copy_(torch.from_numpy(gh))
                elif t.startswith("NEW_X"):
                    if sol.isave[29] >= 30:
                        break
                else:
                    break
            sol.close()
            return trace, mask
        a, mask_on = run(True)
        b, mask_off = run(False)
        assert mask_on == want_mask and mask_off == 0, (p.name, mask_on, mask_off, want_mask)
        assert len(a) == len(b)
        for k, (ra, rb) in enumerate(zip(a, b)):
            assert ra[0] == rb[0] and np.array_equal(ra[1], rb[1]) and ra[3] == rb[3], (p.name, k)
            assert ra[2].tobytes() == rb[2].tobytes() and ra[4].tobytes() == rb[4].tobytes(), (p.name, k)
            assert np.array_equal(ra[5], rb[5]) and ra[6].tobytes() == rb[6].tobytes(), (p.name, k)


def test_ping_pong_entry_with_mirrored_lists_and_across_a_checkpoint():
    """Two corners of the ping-pong entry: (a) a context that mirrors the reference's arrays at every return
    (LBFGSB_F_MIRROR_INDEX: z, d, xp stored, Index / Indx2 kept) driven through the ping-pong entry is bit for bit
    the classic one, exported wa / iwa included; (b) checkpoint / resume: a run exported at a NEW_X return and
    imported into a fresh context continues through the ping-pong entry (iterate in x0, gradient in g0) exactly
    as the uninterrupted run."""
    import numpy as np
    import torch
    import lbfgsb_amd as la
    from oracle import pyoracle as po
    TIME_D = [5, 6, 7, 8, 9]
    for p in (po.problem_quadratic(4099, 7, mixed_nbd=True), po.problem_rosenbrock(1000, 10, 0.0, 0.0)):
        def run(pp, mirror, stop_at=None, resume=None, iters=24):
            sol = la.DeviceSolver(p.n, p.m, mirror_index=mirror)
            x0 = p.x0.copy() if resume is None else resume["x"]
            xs = [torch.from_numpy(x0).cuda(), torch.full((p.n,), 5.0, dtype=torch.float64, device="cuda")]
            gs = [torch.zeros_like(xs[0]), torch.full_like(xs[0], 9.0)]
            if resume is not None:
                gs[0].copy_(torch.from_numpy(resume["g"]))
                sol.import_state(resume["wa"], resume["iwa"], resume["isave"])
                for nm in ("task", "csave", "lsave", "isave", "dsave", "f"):
                    getattr(sol, nm)[:] = resume[nm]
            l, u = torch.from_numpy(p.l).cuda(), torch.from_numpy(p.u).cuda()
            nbd = torch.from_numpy(p.nbd.astype(np.int32)).cuda()
            x, g, trace, saved = xs[0], gs[0], [], None
            for _ in range(400):
                if pp:
                    t, cur = sol.setulb_pp(xs, l, u, nbd, gs, p.factr, p.pgtol)
                    x, g = xs[cur], gs[cur]
                else:
                    t = sol.setulb(x, l, u, nbd, g, p.factr, p.pgtol)
                torch.cuda.synchronize()
                wa, iwa = sol.export_state()
                ds = sol.dsave.copy()
                ds[TIME_D] = 0
                trace.append((t, sol.isave[21:44].copy(), ds, float(sol.f[0]), wa, iwa, x.cpu().numpy()))
                if t.startswith("FG"):
                    xh = x.cpu().numpy()
                    gh = np.empty_like(xh)
                    sol.f[0] = p.fg(xh, gh)
                    g.copy_(torch.from_numpy(gh))
                elif t.startswith("NEW_X"):
                    if stop_at is not None and sol.isave[29] == stop_at:
                        saved = dict(x=x.cpu().numpy(), g=g.cpu().numpy(), wa=wa, iwa=iwa, task=sol.task.copy(),
                                     csave=sol.csave.copy(), lsave=sol.lsave.copy(), isave=sol.isave.copy(),
                                     dsave=sol.dsave.copy(), f=sol.f.copy())
                        break
                    if sol.isave[29] >= iters:
                        break
                else:
                    break
            sol.close()
            return trace, saved

        def same(a, b, what, skip_first=False):
            assert len(a) == len(b), (p.name, what, len(a), len(b))
            for k, (ra, rb) in enumerate(zip(a, b)):
                assert ra[0] == rb[0] and np.array_equal(ra[1], rb[1]) and ra[3] == rb[3], (p.name, what, k)
                assert ra[2].tobytes() == rb[2].tobytes() and ra[6].tobytes() == rb[6].tobytes(), (p.name, what, k)
                if not skip_first:
                    assert ra[4].tobytes() == rb[4].tobytes() and np.array_equal(ra[5], rb[5]), (p.name, what, k)
        # (a) mirrored lists
        a, _ = run(False, True)
        b, _ = run(True, True)
        same(a, b, "mirror")
        # (b) checkpoint at iteration 9, resume through the ping-pong entry
        full, _ = run(True, False)
        head, saved = run(True, False, stop_at=9)
        tail, _ = run(True, False, resume=saved)
        assert saved is not None and len(head) + len(tail) == len(full)
        # (the exported wa of a production context is a VIEW of the state -- z, xp slots are dead storage there --
        #  so across the checkpoint the comparison is on what a caller sees: task, counters, scalars, f, x)
        same(tail, full[len(head):], "resumed", skip_first=True)


def test_contexts_on_four_host_threads_do_not_interfere():
    """include/lbfgsb_hip.h: one host thread per context, any number of contexts.  40 random problems run one after
    the other, then again spread over four host threads that drive their contexts at the same time (ctypes
    releases the GIL inside the library; every context has its own stream): every call of every run bit for
    bit the serial one -- no state shared between contexts."""
    import threading
    import numpy as np
    import torch
    import lbfgsb_amd as la
    from oracle import pyoracle as po
    from test_gpu_fuzz import make

    def run(p, pp, iters=30):
        torch.cuda.set_device(0)
        sol = la.DeviceSolver(p.n, p.m)
        xs = [torch.from_numpy(p.x0.copy()).cuda(), torch.zeros(p.n, dtype=torch.float64, device="cuda")]
        gs = [torch.zeros_like(xs[0]), torch.zeros_like(xs[0])]
        x, g = xs[0], gs[0]
        l, u = torch.from_numpy(p.l).cuda(), torch.from_numpy(p.u).cuda()
        nbd = torch.from_numpy(p.nbd.astype(np.int32)).cuda()
        trace = []
        try:
            for _ in range(100000):
                if pp:
                    t, c = sol.setulb_pp(xs, l, u, nbd, gs, p.factr, p.pgtol)
                    x, g = xs[c], gs[c]
                else:
                    t = sol.setulb(x, l, u, nbd, g, p.factr, p.pgtol)
                ds = sol.dsave.copy()
                ds[[5, 6, 7, 8, 9]] = 0
                trace.append((t, sol.isave[21:44].tobytes(), ds.tobytes(), float(sol.f[0]), x.cpu().numpy().tobytes()))
                if t.startswith("FG"):
                    xh = x.cpu().numpy()
                    gh = np.empty_like(xh)
                    sol.f[0] = p.fg(xh, gh)
                    g.copy_(torch.from_numpy(gh))
                elif t.startswith("NEW_X"):
                    if sol.isave[29] >= iters:
                        break
                else:
                    break
        finally:
            sol.close()
        return trace
    probs = [make(po, s, 3000, 1, 25) for s in range(15000, 15040)]
    serial = [run(p, i & 1) for i, p in enumerate(probs)]
    out, errs = [None] * len(probs), []

    def worker(k):
        try:
            for i in range(k, len(probs), 4):
                out[i] = run(probs[i], i & 1)
        except BaseException as e:   # noqa: BLE001
            errs.append(repr(e))
    ths = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    assert [a == b for a, b in zip(serial, out)] == [True] * len(probs)


def test_null_arguments_are_refused_not_dereferenced():
    """No entry of the C ABI launches a kernel on (or reads through) a null pointer: LBFGSB_E_ARG and a message,
    the context stays usable."""
    import numpy as np
    import torch
    import lbfgsb_amd as la
    from lbfgsb_amd import capi
    lib = la.load_library()
    n, m = 1000, 5
    sol = la.DeviceSolver(n, m)
    try:
        h = sol.h
        x = torch.zeros(n, dtype=torch.float64, device="cuda")
        g = torch.zeros_like(x)
        l, u = torch.full_like(x, -1.0), torch.full_like(x, 1.0)
        nbd = torch.full((n,), 2, dtype=torch.int32, device="cuda")
        P = lambda t: C.c_void_p(t.data_ptr())           # noqa: E731
        A = lambda a: a.ctypes.data_as(C.c_void_p)       # noqa: E731
        task, csave = np.frombuffer(b"START".ljust(60), dtype=np.uint8).copy(), np.zeros(60, np.uint8)
        lsave, isave, dsave, f = np.zeros(4, np.int32), np.zeros(44, np.int32), np.zeros(29), np.zeros(1)
        good = [h, P(x), P(l), P(u), P(nbd), A(f), P(g), 0.0, 0.0, A(task), -1, A(csave), A(lsave), A(isave), A(dsave)]
        for k in (1, 2, 3, 4, 5, 6, 9, 11, 12, 13, 14):
            args = list(good)
            args[k] = None
            assert lib.lbfgsb_hip_setulb_dev(*args) == capi.E_ARG, k
            assert b"NULL" in lib.lbfgsb_hip_last_error()
        out = np.zeros(1)
        assert lib.lbfgsb_hip_projgr(h, None, P(l), P(u), P(nbd), P(g), A(out)) == capi.E_ARG
        assert lib.lbfgsb_hip_wtv(h, None, 1, 1, A(out)) == capi.E_ARG
        assert lib.lbfgsb_hip_objective(h, 0, None, P(g), A(out)) == capi.E_ARG
        assert lib.lbfgsb_hip_export_state(h, None, None) == capi.E_ARG
        assert lib.lbfgsb_hip_import_state(h, None, None, None) == capi.E_ARG
        assert lib.lbfgsb_hip_cauchy(h, None, P(l), P(u), P(nbd), P(g), 1.0, 0, 1, 1.0, None, None, None) == capi.E_ARG
        assert lib.lbfgsb_hip_setulb_host(n, m, None, None, None, None, None, None, 0.0, 0.0, None, None, None, -1,
                                          None, None, None, None, None, 8, 0) == capi.E_ARG
        # a plain host array where a device vector belongs (the classic slip with this entry): refused at START,
        # not a GPU fault
        host = np.zeros(n)
        args = list(good)
        args[1] = A(host)
        assert lib.lbfgsb_hip_setulb_dev(*args) == capi.E_ARG
        assert b"device-accessible" in lib.lbfgsb_hip_last_error()
        # ... and the context still works
        assert sol.setulb(x, l, u, nbd, g, 0.0, 0.0).startswith("FG_START")
    finally:
        sol.close()


def test_bounds_edited_in_place_end_the_run_with_an_error(env):
    """The reference re-reads l, u, nbd on every call (src/lbfgsb.f90:1270-1330, 2594-2622, 2789-2816); the
    passes over W read a snapshot (packed nbd byte; constants or table entries for uniform / few-valued bounds).
    The caller's arrays are compared with the snapshot after the first iteration and then every `bounds_check`
    iterations (default 32): an edit in place is reported -- task 'ERROR: BOUNDS CHANGED DURING RUN' -- instead of
    silently iterated on with stale bounds.  Uniform, dictionary-coded and plain (streamed) bounds; device and
    host form.  (l, u that ARE streamed live take effect as in the reference: only nbd is a snapshot there.)"""
    po, torch, la = env["po"], env["torch"], env["la"]

    def drive(p, edit, every, want_mask, max_iter=40):
        sol = la.DeviceSolver(p.n, p.m, options={"bounds_check": every})
        x = torch.from_numpy(p.x0.copy()).cuda()
        g = torch.zeros_like(x)
        l, u = torch.from_numpy(p.l.copy()).cuda(), torch.from_numpy(p.u.copy()).cuda()
        nbd = torch.from_numpy(p.nbd.astype(np.int32)).cuda()
        t, edited_at = "", None
        for _ in range(10000):
            t = sol.setulb(x, l, u, nbd, g, 0.0, 0.0)
            if t.startswith("FG"):
                xh = x.cpu().numpy()
                gh = np.empty_like(xh)
                sol.f[0] = p.fg(xh, gh)
                g.copy_(torch.from_numpy(gh))
                torch.cuda.synchronize()
            elif t.startswith("NEW_X"):
                if sol.isave[29] == 3 and edited_at is None:
                    assert sol.uniform_bounds() == want_mask
                    edit(l, u, nbd)
                    torch.cuda.synchronize()
                    edited_at = 3
                if sol.isave[29] >= max_iter:
                    break
            else:
                break
        it = int(sol.isave[29])
        sol.close()
        return t, it
    quad = po.problem_quadratic(5003, 6)                       # uniform l, u, nbd: mask 7
    rosen = po.problem_rosenbrock(1000, 8, 0.0, 0.0)           # l alternates: dictionary, mask 11
    mixed = po.problem_quadratic(4099, 6, mixed_nbd=True)      # l, u uniform, nbd streamed: mask 3
    many = po.problem_quadratic(4099, 6)
    many.l[:] = -1.0 - np.arange(many.n) / many.n              # n distinct lower bounds: l streamed live, mask 2 | 4
    def set_u(l, u, nbd): u[7] = 0.5                           # noqa: E306
    def set_l(l, u, nbd): l[11] = -0.25                        # noqa: E306
    def set_nbd(l, u, nbd): nbd[5] = 0                         # noqa: E306
    def nothing(l, u, nbd): pass                               # noqa: E306
    # checked at every NEW_X entry: noticed by the call that follows the edit
    for p, edit, mask in ((quad, set_u, 7), (quad, set_nbd, 7), (rosen, set_l, 11), (rosen, set_nbd, 11),
                          (mixed, set_nbd, 3), (mixed, set_u, 3), (many, set_nbd, 6)):
        t, it = drive(p, edit, 1, mask)
        assert t.startswith("ERROR: BOUNDS CHANGED DURING RUN") and it == 3, (p.name, t, it)
    # a cadence of 16: noticed at the first multiple of 16
    t, it = drive(quad, set_u, 16, 7)
    assert t.startswith("ERROR: BOUNDS CHANGED DURING RUN") and it == 16, (t, it)
    # no edit, or an edit of an array that is streamed live: the run goes on
    t, it = drive(rosen, nothing, 1, 11)
    assert t.startswith("NEW_X") and it == 40
    t, it = drive(many, set_l, 1, 6)
    assert t.startswith("NEW_X") and it == 40
    # switched off: never looked at
    t, it = drive(quad, set_u, 0, 7)
    assert t.startswith("NEW_X") and it == 40
    # the host-pointer form (copies made at START): the caller's arrays are uploaded again and compared
    p = po.problem_quadratic(3001, 5)
    s = po.State.fresh(p)
    nbd = p.nbd.astype(np.int32)
    lo = p.l.copy()
    for _ in range(10000):
        la.setulb(p.n, p.m, s.x, lo, p.u, nbd, s.f, s.g, 0.0, 0.0, s.wa, s.iwa, s.task, -1, s.csave, s.lsave,
                  s.isave, s.dsave)
        if s.task_s.startswith("FG"):
            s.f[0] = p.fg(s.x, s.g)
        elif s.task_s.startswith("NEW_X"):
            if s.isave[29] == 2:
                lo[17] = -0.5
            if s.isave[29] >= 40:
                break
        else:
            break
    assert s.task_s.startswith("ERROR: BOUNDS CHANGED DURING RUN") and s.isave[29] == 32, (s.task_s, s.isave[29])
    assert s.isave[16] == 0 and s.isave[17] == 0         # the context was released with the terminal task


def test_other_bound_arrays_during_a_run_end_the_dictionary_mode_not_the_run(env):
    """Dictionary-coded bounds are a property of the ARRAYS task 'START' looked at: a caller that passes other
    arrays (same contents) on a later call gets the streaming kernels from then on -- same results bit for bit."""
    po, torch, la = env["po"], env["torch"], env["la"]
    p = po.problem_rosenbrock(2000, 7, 0.0, 0.0)

    def run(swap_at):
        sol = la.DeviceSolver(p.n, p.m)
        x = torch.from_numpy(p.x0.copy()).cuda()
        g = torch.zeros_like(x)
        l, u = torch.from_numpy(p.l.copy()).cuda(), torch.from_numpy(p.u.copy()).cuda()
        nbd = torch.from_numpy(p.nbd.astype(np.int32)).cuda()
        l2 = l.clone()
        rows, masks = [], []
        for _ in range(10000):
            t = sol.setulb(x, l, u, nbd, g, 0.0, 0.0)
            masks.append(sol.uniform_bounds())
            if t.startswith("FG"):
                xh = x.cpu().numpy()
                gh = np.empty_like(xh)
                sol.f[0] = p.fg(xh, gh)
                g.copy_(torch.from_numpy(gh))
                torch.cuda.synchronize()
            elif t.startswith("NEW_X"):
                rows.append((tuple(int(v) for v in sol.isave[21:44]), sol.f.tobytes(), x.cpu().numpy().tobytes()))
                if sol.isave[29] == swap_at:
                    l = l2
                if sol.isave[29] >= 25:
                    break
            else:
                break
        sol.close()
        return rows, masks
    a, ma = run(None)
    b, mb = run(6)
    assert a == b
    assert set(ma) == {11} and mb[0] == 11 and mb[-1] == 0

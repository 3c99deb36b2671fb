"""task = 'STOP: CPU ...' (src/lbfgsb.f90:565-573): the caller ends the run and asks for the latest ITERATE back --
x = t, g = r, f = fold -- the contract test/driver3.f90:151-182 relies on (it then prints wa's t slot, dsave(2),
dsave(13)).  With ping-pong iterate buffers, a deferred line-search set-up and lean stores t, r and fold are roles
and pending values, so every entry is checked:

  * one-step parity against the oracle (classic entry, mirroring context): the stop sent at FG_LNSRCH returns
    (first and later trials) and at NEW_X returns of the oracle's trajectory -- every caller array;
  * the production paths (classic / ping-pong / ping-pong + LBFGSB_F_DEFER_LNSRCH with the set-up still deferred
    when the stop arrives): the restored x, g, f must be BIT FOR BIT the iterate the run itself returned at the
    NEW_X before (at a NEW_X return: the iterate before that one, as in the reference, where t still holds it);
  * the host-pointer form: x, g, f and wa(3n + 2mn + 11m^2 + 1 : + n), dsave(2);
  * two ranks (host-callback reducer threads are not needed: each rank restores its own rows -- the sharded
    case is covered through tests/_mr_worker.py in test_gpu_multirank.py::test_stop_cpu_sharded).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STOP = "STOP: CPU EXCEEDING THE TIME LIMIT."


@pytest.fixture(scope="module")
def env(oracle_built):
    import torch
    import lbfgsb_amd
    assert torch.cuda.is_available(), "these tests need the MI355X"
    lbfgsb_amd.load_library()
    return dict(po=oracle_built, torch=torch, la=lbfgsb_amd)


CASES = [("quad1000", dict(kind="quad", n=1000, m=10), 60),
         ("quadmix4099", dict(kind="quadmix", n=4099, m=10), 50),
         ("rosenbrock1000", dict(kind="ros", n=1000, m=10, factr=0.0, pgtol=0.0), 70)]


@pytest.mark.parametrize("name,spec,ncalls", CASES, ids=[c[0] for c in CASES])
def test_stop_cpu_one_step_parity(env, name, spec, ncalls):
    """the oracle's state after return k, task overwritten with the stop, ONE call on both sides"""
    from test_gpu_parity import compare_states, gpu_one_call, make_problem, oracle_snapshots
    po = env["po"]
    p = make_problem(po, spec)
    snaps = oracle_snapshots(po, p, ncalls)
    eng = po.Engine("oracle")
    kinds = {"FG_LN first": 0, "FG_LN later": 0, "NEW_X": 0, "FG_START": 0}
    for k, s in enumerate(snaps):
        t = s.task_s
        if t.startswith("FG_ST"):
            kind = "FG_START"
        elif t.startswith("FG_LN"):
            kind = "FG_LN first" if int(s.isave[35]) == 1 else "FG_LN later"
        elif t.startswith("NEW_X"):
            kind = "NEW_X"
        else:
            continue
        if kinds[kind] >= (6 if kind != "FG_START" else 1):
            continue
        kinds[kind] += 1
        s_in = s.copy()
        s_in.task[:] = po.pad60(STOP)
        exp = s_in.copy()
        po.call(eng, p, exp)
        assert exp.task_s.startswith("STOP: CPU")
        _, out = gpu_one_call(env, p, s_in)
        compare_states(out, exp, p.n, p.m, po)
        # what driver3 prints: the t slot IS the restored x, dsave(2) the restored f
        o, ln = po.wa_offsets(p.n, p.m)["t"]
        assert np.array_equal(out.wa[o:o + ln], out.x)
        assert out.f[0] == out.dsave[1]
    assert kinds["FG_LN first"] >= 4 and kinds["NEW_X"] >= 4, kinds
    if name.startswith("rosen"):
        assert kinds["FG_LN later"] >= 1, kinds


def _drive_until(env, p, mode, stop_at, stop_iter):
    """run p through a production context until the return `stop_at` ('FG' or 'NEW_X') of iteration stop_iter, send
    the stop there; -> (restored x, g, f, iterates recorded at the NEW_X returns, dsave, wa t slot)"""
    torch, la = env["torch"], env["la"]
    pp = mode != "classic"
    defer = mode == "pp_defer"
    sol = la.DeviceSolver(p.n, p.m, defer_lnsrch=defer, same_stream_objective=defer)
    try:
        xs = [torch.from_numpy(p.x0.copy()).cuda(), torch.full((p.n,), 7.0, dtype=torch.float64, device="cuda")]
        gs = [torch.zeros_like(xs[0]), torch.full_like(xs[0], -3.0)]
        l, u = torch.from_numpy(p.l).cuda(), torch.from_numpy(p.u).cuda()
        nbd = torch.from_numpy(p.nbd.astype(np.int32)).cuda()
        x, g = xs[0], gs[0]
        iterates = {}

        def call():
            nonlocal x, g
            if pp:
                t, cur = sol.setulb_pp(xs, l, u, nbd, gs, p.factr, p.pgtol)
                x, g = xs[cur], gs[cur]
                return t
            return sol.setulb(x, l, u, nbd, g, p.factr, p.pgtol)
        f_at = {}
        for _ in range(10000):
            t = call()
            sol.sync()
            it = int(sol.isave[29])
            if t.startswith("FG"):
                if stop_at == "FG" and it == stop_iter and t.startswith("FG_LN"):
                    break
                xh = x.cpu().numpy()
                gh = np.empty_like(xh)
                sol.f[0] = p.fg(xh, gh)
                g.copy_(torch.from_numpy(gh))
                torch.cuda.synchronize()
            elif t.startswith("NEW_X"):
                iterates[it] = (x.cpu().numpy().copy(), g.cpu().numpy().copy(), float(sol.f[0]))
                if stop_at == "NEW_X" and it == stop_iter:
                    break
            else:
                raise AssertionError("run ended early: " + t)
        deferred_before = sol.defer_stats()[0]
        sol.set_task(STOP)
        t = call()
        sol.sync()
        assert t.startswith("STOP: CPU"), t
        wa, _ = sol.export_state()
        from oracle import pyoracle as po
        o, ln = po.wa_offsets(p.n, p.m)["t"]
        return dict(x=x.cpu().numpy(), g=g.cpu().numpy(), f=float(sol.f[0]), iterates=iterates,
                    dsave=sol.dsave.copy(), t_slot=wa[o:o + ln].copy(), deferred=deferred_before,
                    isave=sol.isave.copy())
    finally:
        sol.close()


@pytest.mark.parametrize("mode", ["classic", "pp", "pp_defer"])
@pytest.mark.parametrize("stop_at", ["FG", "NEW_X"])
def test_stop_cpu_production_paths(env, mode, stop_at):
    po = env["po"]
    for p, iters in ((po.problem_quadratic(20011, 7, mixed_nbd=True), (3, 6, 11)),
                     (po.problem_rosenbrock(1000, 10, 0.0, 0.0), (2, 5, 14, 22))):
        for k in iters:
            out = _drive_until(env, p, mode, stop_at, k)
            # an FG_LNSRCH return of iteration k + 1 (isave(30) still says k): t = the iterate of NEW_X k;
            # a NEW_X return k: the line search that produced it started from iterate k - 1, which t still holds
            want = out["iterates"][k if stop_at == "FG" else k - 1] if (k if stop_at == "FG" else k - 1) >= 1 else None
            if want is None:
                continue
            assert np.array_equal(out["x"], want[0]), (mode, stop_at, k, "x is not the previous iterate")
            assert np.array_equal(out["g"], want[1]), (mode, stop_at, k, "g is not the previous gradient")
            assert out["f"] == want[2], (mode, stop_at, k, out["f"], want[2])
            assert out["dsave"][1] == want[2], (mode, stop_at, k, "dsave(2) = fold")
            assert np.array_equal(out["t_slot"], want[0]), (mode, stop_at, k, "wa's t slot")
            if mode == "pp_defer":
                assert out["deferred"] > 0   # (the flag did act in this run)


def test_stop_cpu_host_form(env):
    """the reference's own argument list: what driver3 reads after the stop"""
    po, la = env["po"], env["la"]
    p = po.problem_rosenbrock(1000, 10, 0.0, 0.0)
    eng = po.Engine("oracle")
    for stop_call in (9, 16, 31):
        s = po.State.fresh(p)
        e = po.State.fresh(p)
        nbd = p.nbd.astype(np.int32)
        for k in range(stop_call):
            la.setulb(p.n, p.m, s.x, p.l, p.u, nbd, s.f, s.g, p.factr, p.pgtol, s.wa, s.iwa, s.task, -1, s.csave,
                      s.lsave, s.isave, s.dsave)
            po.call(eng, p, e)
            assert s.task_s == e.task_s
            if s.task_s.startswith("FG"):
                s.f[0] = p.fg(s.x, s.g)
                e.f[0] = p.fg(e.x, e.g)
        assert s.task_s.startswith("FG") or s.task_s.startswith("NEW_X")
        s.task[:] = po.pad60(STOP)
        e.task[:] = po.pad60(STOP)
        la.setulb(p.n, p.m, s.x, p.l, p.u, nbd, s.f, s.g, p.factr, p.pgtol, s.wa, s.iwa, s.task, -1, s.csave,
                  s.lsave, s.isave, s.dsave)
        po.call(eng, p, e)
        assert s.task_s == e.task_s
        o, ln = po.wa_offsets(p.n, p.m)["t"]
        scale = float(np.max(np.abs(e.x)))
        assert np.max(np.abs(s.x - e.x)) <= 1e-10 * scale
        assert np.max(np.abs(s.g - e.g)) <= 1e-10 * max(1.0, float(np.max(np.abs(e.g))))
        assert abs(s.f[0] - e.f[0]) <= 1e-10 * abs(e.f[0])
        assert np.array_equal(s.wa[o:o + ln], s.x)          # driver3.f90:171-175
        assert s.f[0] == s.dsave[1]                          # driver3.f90:181
        assert abs(s.dsave[12] - e.dsave[12]) <= 1e-9 * abs(e.dsave[12])
        assert np.array_equal(s.isave[21:44][[8, 12]], e.isave[21:44][[8, 12]])   # iter, nfgv

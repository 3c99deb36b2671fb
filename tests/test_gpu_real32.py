"""REAL32 context (BASELINE.json configs[4], reference -DREAL32, lbfgsb_kinds_module.F90:29-37):
fp32 storage and kernels, fp64 partial sums and host algebra.  The reference's REAL32 build
does everything in fp32, so agreement is a tolerance sweep, not bit parity:
f must agree with the REAL64 oracle to ~1e-6 and with the REAL32 oracle to fp32 noise."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(oracle_built):
    import torch
    import lbfgsb_amd
    assert torch.cuda.is_available()
    return dict(po=oracle_built, torch=torch, la=lbfgsb_amd)


def test_wtv_fp32_storage_fp64_accumulate(env):
    torch, la = env["torch"], env["la"]
    rng = np.random.default_rng(5)
    for n, m in ((1003, 5), (40001, 10), (8193, 20)):
        ws = rng.standard_normal((m, n)).astype(np.float32)
        wy = rng.standard_normal((m, n)).astype(np.float32)
        v = rng.standard_normal(n).astype(np.float32)
        sol = la.DeviceSolver(n, m, real32=True)
        sol.set_w(ws, wy)
        got = sol.wtv(torch.from_numpy(v).cuda(), m, 1)
        want = np.concatenate([wy.astype(np.float64) @ v.astype(np.float64),
                               ws.astype(np.float64) @ v.astype(np.float64)])
        bound = np.concatenate([np.abs(wy).astype(np.float64) @ np.abs(v),
                                np.abs(ws).astype(np.float64) @ np.abs(v)])
        assert np.all(np.abs(got - want) <= 1e-13 * bound)   # products of fp32 are exact in fp64
        sol.close()


def test_real32_trajectory_tolerance_sweep(env):
    po, torch, la = env["po"], env["torch"], env["la"]
    n, m = 1000, 10
    p64 = po.problem_quadratic(n, m)
    p32 = po.problem_quadratic(n, m, real=np.float32)
    f64 = {}
    po.run(po.Engine("oracle"), p64, max_iter=12,
           snapshot=lambda k, s: f64.__setitem__(int(s.isave[29]), float(s.f[0])) if s.task_s.startswith("NEW_X") else None)
    f32 = {}
    po.run(po.Engine("oracle_r32"), p32, max_iter=12,
           snapshot=lambda k, s: f32.__setitem__(int(s.isave[29]), float(s.f[0])) if s.task_s.startswith("NEW_X") else None)
    # host-pointer form with real_bytes = 4, exactly what the Fortran module passes under -DREAL32
    s = po.State.fresh(p32)
    nbd = p32.nbd.astype(np.int32)
    got = {}
    for _ in range(200):
        la.setulb(n, m, s.x, p32.l, p32.u, nbd, s.f, s.g, 0.0, 0.0, s.wa, s.iwa, s.task, -1,
                  s.csave, s.lsave, s.isave, s.dsave)
        t = s.task_s
        if t.startswith("FG"):
            s.f[0] = p32.fg(s.x, s.g)
        elif t.startswith("NEW_X"):
            got[int(s.isave[29])] = float(s.f[0])
            if s.isave[29] >= 12:
                s.task[:] = po.pad60("STOP: enough")
        else:
            break
    assert s.x.dtype == np.float32 and s.dsave.dtype == np.float32
    assert float(s.dsave[4]) == pytest.approx(1.1920929e-07)        # epsmch of REAL32 (:432)
    assert len(got) >= 8
    for it in sorted(got):
        if it in f64:
            assert got[it] == pytest.approx(f64[it], rel=5e-5), ("vs REAL64 oracle", it)
        if it in f32:
            # the all-fp32 reference is itself ~1e-3 away from the fp64 trajectory (979 Cauchy
            # segments accumulated in fp32 in iteration 1): fp32-noise agreement only
            assert got[it] == pytest.approx(f32[it], rel=5e-3), ("vs REAL32 oracle", it)
    # first iterations are identical decisions: tight agreement there
    for it in (1, 2, 3):
        assert got[it] == pytest.approx(f64[it], rel=2e-6)

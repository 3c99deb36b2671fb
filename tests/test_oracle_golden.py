"""Pin the CPU oracle (oracle/lbfgsb_oracle.c) against golden vectors that the
REAL reference produced (tests/golden/make_golden.py), and, when it is built,
against the reference itself (oracle/_ref).  CPU only."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# dsave(6:10) = cpu1, cachyt, sbtime, lnscht, time1 are wall-clock; isave(24) = itfile unit
TIME_D = [5, 6, 7, 8, 9]


def _mask(name, a):
    a = np.array(a, copy=True)
    if name == "dsave":
        a[..., TIME_D] = 0
    if name == "isave":
        a[..., 23] = 0
    return a


def _problem(po, z, real=np.float64):
    name = str(z["problem"])
    n, m = int(z["n"]), int(z["m"])
    if name == "rosenbrock":
        return po.problem_rosenbrock(n, m, float(z["factr"]), float(z["pgtol"]), real)
    return po.problem_quadratic(n, m, mixed_nbd=name.endswith("mixed"), real=real)


def _stop_rule(case):
    if case.startswith("driver2"):
        lim = 99
    elif case.startswith("driver3"):
        lim = 900
    else:
        return None

    def rule(s):
        if s.isave[33] >= lim:
            return "STOP: TOTAL NO. of f AND g EVALUATIONS EXCEEDS LIMIT"
        if s.dsave[12] <= 1.0e-10 * (1.0 + abs(float(s.f[0]))):
            return "STOP: THE PROJECTED GRADIENT IS SUFFICIENTLY SMALL"
        return None
    return rule


CASES = ["driver1", "driver2", "driver3", "quad1000", "quadmix4096", "driver2_r32",
         "quad1000_r32"]


@pytest.mark.parametrize("case", CASES)
def test_oracle_reproduces_reference_trajectory(oracle_built, case):
    """Whole reverse-communication trajectory, bit for bit (same image, no FMA
    contraction): task, f, x, g, isave, dsave, lsave after every setulb return."""
    po = oracle_built
    z = np.load(os.path.join(GOLD, case + "_traj.npz"))
    r32 = case.endswith("_r32")
    real = np.float32 if r32 else np.float64
    p = _problem(po, z, real)
    eng = po.Engine("oracle_r32" if r32 else "oracle")
    ncalls = z["f"].shape[0]
    xg_calls = set(z["xg_calls"].tolist()) if "xg_calls" in z.files else None
    got = dict(task=[], f=[], x=[], g=[], isave=[], dsave=[], lsave=[])

    def snap(k, s):
        got["task"].append(s.task.copy())
        got["f"].append(s.f[0])
        if xg_calls is None or k in xg_calls:
            got["x"].append(s.x.copy())
            got["g"].append(s.g.copy())
        got["isave"].append(s.isave.copy())
        got["dsave"].append(s.dsave.copy())
        got["lsave"].append(s.lsave.copy())

    po.run(eng, p, max_calls=ncalls, snapshot=snap, on_new_x=_stop_rule(case))
    assert len(got["f"]) == ncalls
    for nm in got:
        a = _mask(nm, np.array(got[nm]))
        b = _mask(nm, z[nm])
        assert a.shape == b.shape, nm
        assert a.tobytes() == b.tobytes(), "%s differs from the reference's golden vector" % nm


@pytest.mark.parametrize("case", ["driver1", "driver3", "quad1000", "quadmix4096"])
def test_oracle_one_step_from_golden_state(oracle_built, case):
    """Load a full reference state (wa, iwa, ...) at return k, evaluate f,g as the
    driver would, make ONE oracle call and compare with the reference's state at
    return k+1 -- every array, bit for bit."""
    po = oracle_built
    z = np.load(os.path.join(GOLD, case + "_state.npz"))
    p = _problem(po, z)
    eng = po.Engine("oracle")
    ks = z["k"].tolist()
    pairs = [(i, i + 1) for i in range(len(ks) - 1) if ks[i + 1] == ks[i] + 1]
    assert pairs
    for i, j in pairs:
        s = po.State(p.n, p.m, z["x"][i].copy(), z["g"][i].copy(), np.array([z["f"][i]]),
                     z["wa"][i].copy(), z["iwa"][i].copy(), z["task"][i].copy(),
                     z["csave"][i].copy(), z["lsave"][i].copy(), z["isave"][i].copy(),
                     z["dsave"][i].copy())
        t = s.task_s
        if t.startswith("FG"):
            s.f[0] = p.fg(s.x, s.g)
        elif not t.startswith("NEW_X"):
            continue
        po.call(eng, p, s)
        for nm in ("x", "g", "wa", "iwa", "task", "csave", "lsave", "isave", "dsave"):
            a = _mask(nm, getattr(s, nm))
            b = _mask(nm, z[nm][j])
            assert a.tobytes() == b.tobytes(), "%s after call %d" % (nm, ks[j])
        assert s.f[0] == z["f"][j]


def test_oracle_matches_reference_transcript_numbers(oracle_built):
    """The reference's own golden transcript (test/OUTPUTS/output_90_1,
    iterate.dat): 23 iterations, 28 evaluations, 47 Cauchy segments, 0 skips,
    per-iteration nseg/nact/itls columns."""
    po = oracle_built
    p = po.problem_rosenbrock(25, 5, 1e7, 1e-5)
    rows = []

    def snap(k, s):
        if s.task_s.startswith("NEW_X"):
            rows.append((int(s.isave[29]), int(s.isave[33]), int(s.isave[32]), int(s.isave[38]),
                         int(s.isave[35]) - 1, float(s.dsave[12]), float(s.f[0])))

    s = po.run(po.Engine("oracle"), p, snapshot=snap)
    assert s.task_s == "CONVERGENCE: REL_REDUCTION_OF_F_<=_FACTR*EPSMCH"
    assert (s.isave[29], s.isave[33], s.isave[21], s.isave[25]) == (23, 28, 47, 0)
    gold = []
    with open(os.path.join(GOLD, "ref_outputs", "iterate.dat")) as fh:
        for line in fh:
            t = line.split()
            if len(t) == 10 and t[0].isdigit() and t[2].isdigit():
                gold.append((int(t[0]), int(t[1]), int(t[2]), int(t[3]), int(t[5]),
                             float(t[8].replace("D", "E")), float(t[9].replace("D", "E"))))
    assert len(gold) == 23 and len(rows) == 23
    for a, b in zip(rows, gold):
        assert a[:5] == b[:5]
        assert a[5] == pytest.approx(b[5], rel=2e-3)
        assert a[6] == pytest.approx(b[6], rel=2e-3)


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref",
                                                    "liblbfgsb_ref.so")),
                    reason="oracle/_ref not built (no reference sources here)")
def test_oracle_bit_identical_to_live_reference(oracle_built):
    """A size and (n, m) that are NOT in the fixtures, against the live reference."""
    po = oracle_built
    for p in (po.problem_quadratic(3000, 7, mixed_nbd=True), po.problem_rosenbrock(501, 4, 0.0, 0.0)):
        a, b = [], []
        po.run(po.Engine("oracle"), p, max_calls=120, snapshot=lambda k, s: a.append(s.copy()))
        po.run(po.Engine("ref"), p, max_calls=120, snapshot=lambda k, s: b.append(s.copy()))
        assert len(a) == len(b)
        for sa, sb in zip(a, b):
            for nm in ("x", "g", "f", "wa", "iwa", "task", "csave", "lsave", "isave", "dsave"):
                assert _mask(nm, getattr(sa, nm)).tobytes() == _mask(nm, getattr(sb, nm)).tobytes(), nm

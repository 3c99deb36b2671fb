#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REAL reference.

Run in the build container (needs oracle/_ref/liblbfgsb_ref*.so, i.e.
/root/reference + amdflang):

    make -C oracle all && python tests/golden/make_golden.py

Every fixture is DATA: inputs and the outputs the untouched reference
(jacobwilliams/lbfgsb, built by oracle/Makefile) produced for them.  The
objective values fed to the reference come from oracle/lbfgsb_oracle.c's
lbo_rosenbrock_fg (formulas of reference test/driver1.f90:274-289) and
lbo_quadratic_fg (problem defined in SURVEY.md 8d).

Files
  <name>_traj.npz   per setulb return k: task, f, x, g, isave, dsave, lsave
  <name>_state.npz  full caller state (wa, iwa, csave too) at selected returns;
                    consecutive pairs (k, k+1) allow one-step parity tests.
  ref_outputs/      the reference's own golden transcripts (test/OUTPUTS/*).
"""
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def driver3_stop(s):
    """test/driver3.f90:219-226 user stops."""
    if s.isave[33] >= 900:
        return "STOP: TOTAL NO. of f AND g EVALUATIONS EXCEEDS LIMIT"
    if s.dsave[12] <= 1.0e-10 * (1.0 + abs(float(s.f[0]))):
        return "STOP: THE PROJECTED GRADIENT IS SUFFICIENTLY SMALL"
    return None


def driver2_stop(s):
    """test/driver2.f90:174-185 user stops."""
    if s.isave[33] >= 99:
        return "STOP: TOTAL NO. of f AND g EVALUATIONS EXCEEDS LIMIT"
    if s.dsave[12] <= 1.0e-10 * (1.0 + abs(float(s.f[0]))):
        return "STOP: THE PROJECTED GRADIENT IS SUFFICIENTLY SMALL"
    return None


def transcripts():
    """The reference's debugging output (iprint = 99, 100, 101) on two small problems, captured
    from its stdout: expected text for the product's report layer (src/lbfgsb.f90 iprint >= 99)."""
    import subprocess
    worker = os.path.join(ROOT, "tests", "_iprint_worker.py")
    for problem, n, m, ipr, iters in (("rosen", 7, 5, 101, 6), ("quadmix", 12, 4, 100, 6),
                                      ("quadmix", 12, 4, 99, 6)):
        out = subprocess.run([sys.executable, worker, "ref", problem, str(n), str(m), str(ipr),
                              str(iters)], capture_output=True, text=True, check=True).stdout
        path = os.path.join(OUT, "ref_outputs", "iprint%d_%s_n%d_m%d.txt" % (ipr, problem, n, m))
        with open(path, "w") as fh:
            fh.write(out)
        print("wrote", path, len(out.splitlines()), "lines")


def record(kind, name, p, full_calls, on_new_x=None, max_calls=10**9, lite_xg=None):
    eng = po.Engine(kind)
    traj = dict(task=[], f=[], x=[], g=[], isave=[], dsave=[], lsave=[])
    full = dict(k=[], wa=[], iwa=[], csave=[], x=[], g=[], f=[], task=[], isave=[], dsave=[],
                lsave=[])

    def snap(k, s):
        traj["task"].append(s.task.copy())
        traj["f"].append(s.f[0])
        if lite_xg is None or k in lite_xg:
            traj["x"].append(s.x.copy())
            traj["g"].append(s.g.copy())
        traj["isave"].append(s.isave.copy())
        traj["dsave"].append(s.dsave.copy())
        traj["lsave"].append(s.lsave.copy())
        if full_calls == "all" or k in full_calls:
            full["k"].append(k)
            for nm in ("wa", "iwa", "csave", "x", "g", "task", "isave", "dsave", "lsave"):
                full[nm].append(getattr(s, nm).copy())
            full["f"].append(s.f[0])

    s = po.run(eng, p, max_calls=max_calls, snapshot=snap, on_new_x=on_new_x)
    meta = dict(n=p.n, m=p.m, factr=p.factr, pgtol=p.pgtol, problem=p.name,
                x0=p.x0, l=p.l, u=p.u, nbd=p.nbd)
    if lite_xg is not None:
        meta["xg_calls"] = np.array(sorted(lite_xg))
    np.savez_compressed(os.path.join(OUT, name + "_traj.npz"),
                        **{k: np.array(v) for k, v in traj.items()}, **meta)
    if full["k"]:
        np.savez_compressed(os.path.join(OUT, name + "_state.npz"),
                            **{k: np.array(v) for k, v in full.items()}, **meta)
    print("%-22s calls=%d iters=%d final task=%s f=%.17g" % (
        name, len(traj["f"]), s.isave[29], s.task_s, float(s.f[0])))


def main():
    # (i) driver1: n=25 m=5 factr=1e7 pgtol=1e-5, everything at every return
    record("ref", "driver1", po.problem_rosenbrock(25, 5, 1e7, 1e-5), "all")
    # driver2: n=25 m=5 factr=pgtol=0 with the user stops
    record("ref", "driver2", po.problem_rosenbrock(25, 5, 0.0, 0.0), "all", driver2_stop)
    # (ii) driver3: n=1000 m=10, user stops; full state at a few call pairs
    record("ref", "driver3", po.problem_rosenbrock(1000, 10, 0.0, 0.0),
           {0, 1, 2, 5, 6, 7, 12, 13, 30, 31, 60, 61}, driver3_stop)
    # (iii) separable quadratic n=1000 m=10 (runs to REL_REDUCTION convergence)
    record("ref", "quad1000", po.problem_quadratic(1000, 10),
           {0, 1, 2, 3, 4, 5, 20, 21, 22, 23, 80, 81, 149, 150})
    # mixed bound types nbd = mod(i,4), n=4096: trajectory scalars + x,g at a few returns
    record("ref", "quadmix4096", po.problem_quadratic(4096, 10, mixed_nbd=True),
           {3, 4, 40, 41}, lite_xg={0, 1, 2, 3, 4, 40, 41, 100, 178})
    # (iv) REAL32 build of the reference, driver2 settings
    if po.Engine.available("ref_r32"):
        record("ref_r32", "driver2_r32",
               po.problem_rosenbrock(25, 5, 0.0, 0.0, real=np.float32), "all", driver2_stop)
        record("ref_r32", "quad1000_r32", po.problem_quadratic(1000, 10, real=np.float32),
               {0, 1, 2, 3, 10, 11}, max_calls=60)
    # the reference's own golden transcripts are data files of its test suite
    src = "/root/reference/test/OUTPUTS"
    if os.path.isdir(src):
        dst = os.path.join(OUT, "ref_outputs")
        os.makedirs(dst, exist_ok=True)
        for fn in ("output_90_1", "output_90_2", "output_90_3", "iterate.dat"):
            shutil.copyfile(os.path.join(src, fn), os.path.join(dst, fn))


if __name__ == "__main__":
    if sys.argv[1:] == ["transcripts"]:
        transcripts()
    else:
        main()
        transcripts()

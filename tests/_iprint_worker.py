"""Print the transcript of a short run at a given iprint (the reference's debugging output,
src/lbfgsb.f90 iprint >= 99).  engine 'ref' = the real reference (oracle/_ref; used by
tests/golden/make_golden.py in the build container only), 'gpu' = the product through the
reference-shaped host entry.  problem 'rosen': driver1's Rosenbrock; 'quadmix': the bounded
quadratic with all four bound types (variables leave and enter the free set); 'fuzz:SEED:NMAX:MLO:MHI' /
'fam:NAME:SEED': a random problem of tests/test_gpu_fuzz.py (n, m arguments ignored) -- on the GPU box both
engines can run side by side (oracle/_ref travels there as a built library)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(engine, problem, n, m, iprint, iters):
    from oracle import pyoracle as po
    if problem.startswith("fuzz:"):        # fuzz:SEED:NMAX:MLO:MHI -- a random problem of tests/test_gpu_fuzz.py
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import test_gpu_fuzz as tf
        a = problem.split(":")
        p = tf.make(po, int(a[1]), int(a[2]), int(a[3]), int(a[4]))
    elif problem.startswith("fam:"):       # fam:NAME:SEED -- one of its FAMILIES
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import test_gpu_fuzz as tf
        a = problem.split(":")
        p = tf.FAMILIES[a[1]](po, int(a[2]))
    else:
        p = po.problem_rosenbrock(n, m) if problem == "rosen" else po.problem_quadratic(n, m, mixed_nbd=True)
    if engine == "ref":
        e = po.Engine("ref")
        s = po.State.fresh(p, e.int)
        step = lambda: po.call(e, p, s, iprint=iprint)   # noqa: E731
    else:
        import lbfgsb_amd as la
        s = po.State.fresh(p)
        nbd = p.nbd.astype(np.int32)
        step = lambda: la.setulb(p.n, p.m, s.x, p.l, p.u, nbd, s.f, s.g, p.factr, p.pgtol, s.wa,  # noqa: E731
                                 s.iwa, s.task, iprint, s.csave, s.lsave, s.isave, s.dsave)
    for _ in range(100000):
        step()
        t = s.task_s
        if t.startswith("FG"):
            s.f[0] = p.fg(s.x, s.g)
        elif t.startswith("NEW_X"):
            if s.isave[29] >= iters:
                s.task[:] = po.pad60("STOP: ITERATION LIMIT OF THE TRANSCRIPT TEST")
        else:
            break
    sys.stdout.flush()


if __name__ == "__main__":
    a = sys.argv
    main(a[1], a[2], int(a[3]), int(a[4]), int(a[5]), int(a[6]))

#!/usr/bin/env python3
"""Anchors for BASELINE.json configs[2] at its stated size: extended Rosenbrock with box bounds
(test/driver3.f90:102-120, 189-204), n = 1e7, m = 10, fp64 -- per-iteration (iter, nfg, nseg,
nfree, f) produced by the REAL reference (oracle/_ref/liblbfgsb_ref.so, built by oracle/Makefile
from /root/reference), run in the build container:

    python tests/golden/make_anchors_n1e7.py     ->  tests/golden/rosenbrock_n1e7_anchors.json

The objective values fed to the reference come from oracle/lbfgsb_oracle.c's lbo_rosenbrock_fg
(formulas of reference test/driver1.f90:274-289).  The fixture is data: ~1 KB of numbers."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

n, m, iters = 10_000_000, 10, 16
p = po.problem_rosenbrock(n, m, 0.0, 0.0)
rows = []
t0 = time.time()


def snap(k, s):
    if s.task_s.startswith("NEW_X"):
        rows.append(dict(iter=int(s.isave[29]), nfg=int(s.isave[33]), nseg=int(s.isave[32]),
                         nfree=int(s.isave[37]), f=float(s.f[0]), sbgnrm=float(s.dsave[12])))
        print(rows[-1], "%.1f s" % (time.time() - t0), flush=True)


po.run(po.Engine("ref"), p, max_iter=iters, snapshot=snap)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rosenbrock_n1e7_anchors.json")
json.dump(dict(problem="extended Rosenbrock, driver3 bounds, x0 = 3", n=n, m=m, engine="reference (amdflang -O2)",
               rows=rows), open(out, "w"), indent=1)

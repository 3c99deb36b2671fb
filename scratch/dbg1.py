import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import lbfgsb_amd
from oracle import pyoracle as po
p = po.problem_rosenbrock(25, 5, 1e7, 1e-5)
s = po.State.fresh(p); nbd = p.nbd.astype(np.int32)
for k in range(3):
    lbfgsb_amd.setulb(p.n, p.m, s.x, p.l, p.u, nbd, s.f, s.g, p.factr, p.pgtol, s.wa, s.iwa, s.task, -1, s.csave, s.lsave, s.isave, s.dsave, mirror=True)
    print(k, s.task_s, s.x[:6], 'nseg', s.isave[32], 'nfree', s.isave[37])
    if s.task_s.startswith('FG'): s.f[0] = p.fg(s.x, s.g)

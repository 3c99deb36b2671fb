"""The N>1 path on the one GPU a test box has: (a) two ranks sharing cuda:0 with a gloo
host group completing the reductions and the breakpoint all-gather, (b) an RCCL
communicator of ONE rank, which drives ncclAllReduce/ncclAllGather on the solver's stream.
Both must reproduce the single-rank oracle trajectory: integers exactly, f to 1e-9."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(world, mode, n, m, iters, mixed, out):
    port = free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_mr_worker.py"), str(r), str(world),
                               str(port), mode, str(n), str(m), str(iters),
                               mixed if isinstance(mixed, str) else ("1" if mixed else "0"), out])
             for r in range(world)]
    rcs = [p.wait(timeout=300) for p in procs]
    assert rcs == [0] * world, rcs
    return json.load(open(out))


def oracle_rows(po, n, m, iters, mixed):
    if mixed == "rosen":
        p = po.problem_rosenbrock(n, m, factr=0.0, pgtol=0.0)
    else:
        p = po.problem_quadratic(n, m, mixed_nbd=(mixed is True))
    rows = []
    s = po.run(po.Engine("oracle"), p, max_iter=iters,
               snapshot=lambda k, s: rows.append([int(s.isave[29]), int(s.isave[33]), int(s.isave[32]),
                                                  int(s.isave[37]), float(s.f[0]), float(s.dsave[12])])
               if s.task_s.startswith("NEW_X") else None)
    return rows, s.x.copy()


@pytest.mark.parametrize("world,mode,n,m,iters,mixed", [
    (2, "gloo", 20011, 7, 8, True),       # ragged split, all four bound types
    (3, "gloo", 300000, 10, 3, False),    # iteration 1 walks ~293k breakpoints: full sort + merged chunks
    (4, "gloo", 10007, 5, 6, True),       # four ranks on one GPU, m = 5
    (1, "rccl1", 50021, 5, 6, True),
])
def test_sharded_trajectory_matches_oracle(oracle_built, tmp_path, world, mode, n, m, iters, mixed):
    po = oracle_built
    res = launch(world, mode, n, m, iters, mixed, str(tmp_path / "out.json"))
    rows, x = oracle_rows(po, n, m, iters, mixed)
    assert len(res["rows"]) == len(rows) == iters
    for a, b in zip(res["rows"], rows):
        assert a[:4] == b[:4], (a, b)                      # iter nfgv nseg nfree
        assert a[4] == pytest.approx(b[4], rel=1e-9)       # f
        assert a[5] == pytest.approx(b[5], rel=1e-7)       # |proj g|
    xa = np.array(res["x"])
    assert np.max(np.abs(xa - x)) <= 1e-8 * max(1.0, np.max(np.abs(x)))
    if n >= 300000:
        assert res["stats"]["cauchy_fullsorts"] >= 1


def test_sharded_rosenbrock_halo_objective(oracle_built, tmp_path):
    """Extended Rosenbrock (test/driver1.f90:272-291 formulas, driver bounds) with the rows cut
    over 3 ranks: the built-in objective needs x(i-1), x(i+1) across the cuts (1-element halo,
    all-gathered).  Same trajectory as the single-rank oracle."""
    po = oracle_built
    n, m, iters = 30011, 5, 10
    res = launch(3, "gloo", n, m, iters, "rosen", str(tmp_path / "out.json"))
    rows, x = oracle_rows(po, n, m, iters, "rosen")
    assert len(res["rows"]) == len(rows) == iters
    for a, b in zip(res["rows"], rows):
        assert a[:4] == b[:4], (a, b)
        assert a[4] == pytest.approx(b[4], rel=1e-9)
    xa = np.array(res["x"])
    assert np.max(np.abs(xa - x)) <= 1e-8 * max(1.0, np.max(np.abs(x)))


def test_sharded_parallel_gcp(oracle_built, tmp_path):
    """The opt-in closed-form GCP with the rows cut over 2 ranks (one summed count, no
    breakpoint exchange): nseg within 2 of the oracle, f to 1e-9, no sort anywhere."""
    po = oracle_built
    n, m, iters = 200003, 10, 3
    res = launch(2, "gloo", n, m, iters, "pgcp", str(tmp_path / "out.json"))
    rows, x = oracle_rows(po, n, m, iters, False)
    assert len(res["rows"]) == len(rows) == iters
    for a, b in zip(res["rows"], rows):
        assert a[:2] == b[:2], (a, b)
        assert abs(a[2] - b[2]) <= 2 and abs(a[3] - b[3]) <= 2, (a, b)
        assert a[4] == pytest.approx(b[4], rel=1e-9)
    assert res["stats"]["cauchy_fullsorts"] == 0

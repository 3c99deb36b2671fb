"""CPU (gloo, world_size 2) coverage of the host side of the N>1 path: the row partition
and the group reducers that complete sum|min|max partials and gather breakpoint chunks."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_block_partition_covers_rows():
    from lbfgsb_amd.distributed import block_partition
    for n in (1, 7, 8, 1000003, 10**8):
        for world in (1, 2, 3, 8):
            if n < world:
                continue
            nxt = 0
            for r in range(world):
                row0, nl = block_partition(n, world, r)
                assert row0 == nxt and nl >= n // world
                nxt += nl
            assert nxt == n
    with pytest.raises(ValueError):
        block_partition(10, 2, 2)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from lbfgsb_amd.distributed import make_group_reducers
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    ar, ag = make_group_reducers()
    # layout of a reduction buffer: nsum sums | nmin mins | nmax maxes
    buf = np.array([1.0 + rank, 10.0 * (rank + 1), 5.0 - rank, 7.0 + 3 * rank, -2.0 * rank], np.float64)
    ar(buf, 2, 2, 1)
    g = ag(np.arange(4, dtype=np.uint8) + 10 * rank)
    q.put((rank, buf.tolist(), g.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_group_reducers_gloo_world2():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    got = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in ps]
    for rank, buf, g in got:
        assert buf == [3.0, 30.0, 4.0, 7.0, 0.0]           # sums, sums, min, min, max
        assert g == [0, 1, 2, 3, 10, 11, 12, 13]           # rank-major gather


def test_c_abi_exports_every_declared_symbol():
    """include/*.h (the drop-in surface lbfgsb_hip.h + the instruments of lbfgsb_hip_debug.h) vs the shared
    library: every declared entry point resolves (no compute call is made -- this runs without a GPU)."""
    import re
    import lbfgsb_amd
    from lbfgsb_amd import capi
    import glob
    hdr = "".join(open(f).read() for f in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))))
    declared = set(re.findall(r"\b(lbfgsb_hip_[a-z0-9_]+)\s*\(", hdr)) - {"lbfgsb_hip_ctx"}
    assert declared == set(capi.PROTOTYPES), declared ^ set(capi.PROTOTYPES)
    lib = lbfgsb_amd.load_library()
    for name in declared:
        assert getattr(lib, name) is not None


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    import lbfgsb_amd
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(lbfgsb_amd.LbfgsbError, match="no HIP device|no CPU path"):
        lbfgsb_amd.DeviceSolver(100, 5)
    x = np.zeros(10)
    with pytest.raises(lbfgsb_amd.LbfgsbError):
        from oracle import pyoracle as po
        p = po.problem_quadratic(10, 3)
        s = po.State.fresh(p)
        lbfgsb_amd.setulb(10, 3, s.x, p.l, p.u, p.nbd, s.f, s.g, 0.0, 0.0, s.wa, s.iwa, s.task, -1,
                          s.csave, s.lsave, s.isave, s.dsave)


def test_header_and_c_example_are_plain_c():
    """include/lbfgsb_hip.h is the boundary a C caller compiles against: it and the plain-C twin of
    the reference's driver1 (examples/driver1.c) must be valid C99 (no C++-isms in the header)."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cc = shutil.which("gcc") or shutil.which("cc")
    assert cc, "no C compiler"
    subprocess.check_call([cc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                           "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "driver1.c")])


def test_bench_parity_in_run_flags_what_it_should():
    """bench.py's in-run check (the timed leg's NEW_X rows against the rows the real reference printed at full
    size): the fixture's own rows pass, a changed integer or an f off by 1e-8 fails, other shapes are not compared."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "quad_n1e8_m10_ref_rows.json")))["rows"]
    rows = [(r["iter"], r["nfg"], r["nseg"], r["nfree"], r["f"]) for r in ref]
    more = rows + [(17, 19, 1, 49999496, 1.0), (18, 20, 1, 49999496, 0.9)]      # (iterations beyond the fixture)
    ok = bench.parity_in_run(more, 100_000_000, 10, False, 0)
    assert ok["ok"] is True and ok["rows_checked"] == len(ref)
    bad = list(rows)
    bad[3] = (bad[3][0], bad[3][1], bad[3][2] + 1, bad[3][3], bad[3][4])
    r = bench.parity_in_run(bad, 100_000_000, 10, False, 0)
    assert r["ok"] is False and r["mismatch"][0]["got"][0] == 4
    bad = list(rows)
    bad[7] = bad[7][:4] + (bad[7][4] * (1 + 1e-8),)
    assert bench.parity_in_run(bad, 100_000_000, 10, False, 0)["ok"] is False
    assert bench.parity_in_run(rows, 3_000_000, 10, False, 0)["ok"] is None          # no rows on file for this shape
    r6 = json.load(open(os.path.join(ROOT, "tests", "golden", "quad_n1e6_m10_ref_rows.json")))["rows"]
    assert bench.parity_in_run([(r["iter"], r["nfg"], r["nseg"], r["nfree"], r["f"]) for r in r6], 1_000_000, 10,
                               False, 0)["ok"] is True                                 # configs[1] has rows too
    # REAL32 contexts are compared with the REAL64 reference's rows of the same shape under REAL32 rules
    r20 = json.load(open(os.path.join(ROOT, "tests", "golden", "quad_n1e8_m20_ref_rows.json")))["rows"]
    near = [(r["iter"], r["nfg"], r["nseg"] + 3, r["nfree"] - 4, r["f"] * (1 + 5e-7)) for r in r20]
    assert bench.parity_in_run(near, 100_000_000, 20, True, 0)["ok"] is True
    far = [(r["iter"], r["nfg"] + (r["iter"] == 20), r["nseg"], r["nfree"], r["f"]) for r in r20]
    assert bench.parity_in_run(far, 100_000_000, 20, True, 0)["ok"] is False
    assert bench.parity_in_run(rows, 100_000_000, 10, True, 0)["ok"] is None          # REAL32: not the fixture's kind
    assert bench.parity_in_run([], 100_000_000, 10, False, 0)["ok"] is False         # nothing ran: not a pass

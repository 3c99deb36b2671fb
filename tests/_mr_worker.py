"""Worker for the multi-rank tests: `world` processes, one shard of the rows each.
mode 'gloo': ranks share cuda:0, reductions through a gloo host group (host callbacks).
mode 'rccl1': world must be 1; an RCCL communicator of one rank is attached so that the
ncclAllReduce / ncclAllGather code path runs on a single GPU.
mode 'fakerccl': several ranks on cuda:0 through the library's communicator code path, with a
shared-memory stand-in for librccl (tests/fake_rccl.cpp, LBFGSB_RCCL_LIBRARY)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(rank, world, port, mode, n, m, iters, variant, out_path):
    # variant: "0" quadratic | "1" quadratic with all four bound types | "rosen" box-bounded
    # extended Rosenbrock (1-element halo in the objective) | "pgcp" quadratic with the
    # opt-in closed-form GCP
    import torch
    import torch.distributed as dist
    import lbfgsb_amd
    from oracle import pyoracle as po

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank,
                            world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    row0, n_loc = lbfgsb_amd.block_partition(n, world, rank)
    def make_solver():
        s_ = lbfgsb_amd.DeviceSolver(n_loc, m, n_global=n, row0=row0, device=0,
                                     parallel_gcp=(variant in ("pgcp", "pgcp2")),
                                     # LBFGSB_F_DEFER_LNSRCH over several ranks (the deferred sums are one more
                                     # reduced segment of every rank's fetch): tests/test_gpu_multirank.py
                                     defer_lnsrch=os.environ.get("LBFGSB_TEST_DEFER") == "1",
                                     # (this worker evaluates f, g with sol.objective: the solver's own stream)
                                     same_stream_objective=os.environ.get("LBFGSB_TEST_DEFER") == "1",
                                     index_ties=(variant == "sym"),
                                     options=({"exact_always": 1} if variant == "symx" else
                                              # the two passes over W on the tile-local free-row layout, re-sorted in
                                              # every iteration (tests/test_gpu_compact.py)
                                              {"compact_w": 2, "compact_policy": 2}
                                              if os.environ.get("LBFGSB_TEST_COMPACT") == "1" else None))
        if mode == "gloo":
            lbfgsb_amd.attach_host_group(s_, rank, world)
        elif mode == "rccl1":
            assert world == 1
            lbfgsb_amd.attach_rccl(s_, 0, 1, dev)
        elif mode == "fakerccl":
            # the library's communicator code path (ncclAllReduce / ncclAllGather on the solver's
            # stream) with several ranks on ONE GPU: LBFGSB_RCCL_LIBRARY points at tests/fake_rccl.cpp
            ids = [lbfgsb_amd.DeviceSolver.rccl_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, 0)
            s_.init_rccl(ids[0], rank, world)
            assert "libfake_rccl" in open("/proc/self/maps").read()   # (not the real library)
        return s_
    sol = make_solver()
    two_scale = None
    if variant == "pgcp2":
        p, two_scale = two_scale_problem(po, n, m)
    elif variant in ("sym", "symx"):
        p, two_scale = symmetric_problem(po, n, m)
    elif variant == "rosen":
        p = po.problem_rosenbrock(n, m, factr=0.0, pgtol=0.0)
    else:
        p = po.problem_quadratic(n, m, mixed_nbd=(variant == "1"))
    kind = 1 if variant == "rosen" else 0
    sl = slice(row0, row0 + n_loc)
    x = torch.from_numpy(p.x0[sl].copy()).to(dev)
    g = torch.zeros_like(x)
    l = torch.from_numpy(p.l[sl].copy()).to(dev)
    u = torch.from_numpy(p.u[sl].copy()).to(dev)
    nbd = torch.from_numpy(p.nbd[sl].astype(np.int32)).to(dev)
    rows = []
    stopcpu = None
    for _ in range(100000):
        t = sol.setulb(x, l, u, nbd, g, 0.0, 0.0)
        if variant == "stopcpu" and t.startswith("FG_LN") and sol.isave[29] == iters:
            # 'STOP: CPU' (src/lbfgsb.f90:565-573) at the first trial point after iteration `iters`: every rank gets
            # ITS rows of the iterate back, bit for bit, and the global f
            sol.sync()
            sol.set_task("STOP: CPU EXCEEDING THE TIME LIMIT.")
            t = sol.setulb(x, l, u, nbd, g, 0.0, 0.0)
            sol.sync()
            ok = (t.startswith("STOP: CPU") and bool(torch.equal(x, stopcpu[0])) and bool(torch.equal(g, stopcpu[1]))
                  and float(sol.f[0]) == stopcpu[2] and float(sol.dsave[1]) == stopcpu[2])
            oks = [None] * world
            dist.all_gather_object(oks, ok)
            stopcpu = all(oks)
            break
        if t.startswith("FG") and two_scale is not None:
            # host objective on this rank's rows, f summed over the ranks
            xh = x.cpu().numpy()
            gh = np.empty_like(xh)
            ft = torch.tensor([two_scale(xh, gh, row0, row0 + n_loc)], dtype=torch.float64)
            dist.all_reduce(ft)
            sol.f[0] = float(ft[0])
            g.copy_(torch.from_numpy(gh))
        elif t.startswith("FG"):
            sol.objective(kind, x, g, deferred=True)   # global f: fetched by the next setulb call
        elif t.startswith("NEW_X"):
            rows.append([int(sol.isave[29]), int(sol.isave[33]), int(sol.isave[32]),
                         int(sol.isave[37]), float(sol.f[0]), float(sol.dsave[12])])
            if variant == "stopcpu" and sol.isave[29] == iters:
                sol.sync()
                stopcpu = (x.clone(), g.clone(), float(sol.f[0]))
                continue
            if sol.isave[29] >= iters:
                break
            if variant == "ckpt" and sol.isave[29] == iters // 2:
                # checkpoint / resume of a SHARDED run: every rank exports its rows in the
                # reference's wa / iwa layout, the context is destroyed, a new one imports the
                # state and carries on from the saved task / csave / lsave / isave / dsave
                wa, iwa = sol.export_state()
                saved = (sol.task.copy(), sol.csave.copy(), sol.lsave.copy(), sol.isave.copy(),
                         sol.dsave.copy(), sol.f.copy())
                sol.close()
                dist.barrier()
                sol = make_solver()
                sol.import_state(wa, iwa, saved[3])
                sol.task[:], sol.csave[:], sol.lsave[:] = saved[0], saved[1], saved[2]
                sol.isave[:], sol.dsave[:], sol.f[:] = saved[3], saved[4], saved[5]
        else:
            break
    torch.cuda.synchronize()
    xs = [None] * world
    dist.all_gather_object(xs, x.cpu().numpy())
    if rank == 0:
        with open(out_path, "w") as fh:
            json.dump({"rows": rows, "task": sol.task_s, "x": np.concatenate(xs).tolist(),
                       "stats": sol.stats(), "tie_splits": sol.tie_splits(), "defer": list(sol.defer_stats()),
                       "stopcpu": stopcpu if isinstance(stopcpu, bool) else None}, fh)
    sol.close()
    dist.barrier()
    dist.destroy_process_group()


def two_scale_problem(po, n, m):
    """Separable quadratic with two curvature scales: iterations 2 and 6 cross ~ n/2 breakpoints
    with pairs stored (the col > 0 case of the parallel GCP search)."""
    rng = np.random.default_rng(7)
    a = 1.0 + 99.0 * rng.random(n)
    a[n // 2:] *= 1e-4
    c = rng.choice([-1.0, 1.0], n) * 5.0 * (1.0 + rng.random(n))
    eps = 1e-3

    def fg(x, g, lo=0, hi=None):
        hi = n if hi is None else hi
        d = x - c[lo:hi]
        g[:] = eps * a[lo:hi] * d
        return float(0.5 * eps * np.sum(a[lo:hi] * d * d))
    p = po.Problem("two_scale", n, m, np.zeros(n), -np.ones(n), np.ones(n), np.full(n, 2, np.int32),
                   0.0, 0.0, fg, np.float64)
    return p, fg


def symmetric_problem(po, n, m):
    """Separable quadratic made of 8 exact copies of the same n/8 variables (copy k of variable b
    is row k * (n/8) + b): whole groups of EQUAL breakpoints stay alive over many iterations, and
    the members of a group sit on different ranks."""
    copies = 8
    base = n // copies
    assert base * copies == n
    b = np.arange(n) % base
    a = 1.0 + 99.0 * ((7919 * (b + 1)) % 10007) / 10006.0
    c = -2.0 + 4.0 * ((104729 * (b + 1)) % 100003) / 100002.0

    def fg(x, g, lo=0, hi=None):
        hi = n if hi is None else hi
        d = x - c[lo:hi]
        g[:] = a[lo:hi] * d
        return float(0.5 * np.sum(a[lo:hi] * d * d))
    p = po.Problem("sym_quadratic", n, m, np.zeros(n), -np.ones(n), np.ones(n), np.full(n, 2, np.int32),
                   0.0, 0.0, fg, np.float64)
    return p, fg


def fuzz_problem(po, seed):
    """Random separable box-constrained problem (all bound types, fixed and unbounded variables,
    optional non-convex term), shared by the sharded fuzz run and its single-rank oracle."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(3, 2000))
    m = int(rng.integers(1, 13))
    a = 1.0 + 99.0 * rng.random(n)
    c = rng.normal(0, 2, n)
    wavy = bool(rng.random() < 0.4)

    def fg(x, g, lo=0, hi=None):
        hi = n if hi is None else hi
        d = x - c[lo:hi]
        f = 0.5 * np.sum(a[lo:hi] * d * d)
        g[:] = a[lo:hi] * d
        if wavy:
            f += np.sum(np.cos(3 * x))
            g[:] -= 3 * np.sin(3 * x)
        return float(f)
    l = rng.normal(-1, 1, n)
    u = l + np.abs(rng.normal(1.5, 1, n))
    fixed = rng.random(n) < 0.03
    u[fixed] = l[fixed]
    nbd = rng.integers(0, 4, n).astype(np.int32)
    x0 = rng.normal(0, 3, n)
    return po.Problem("mrfuzz%d" % seed, n, m, x0, l, u, nbd, 0.0, 0.0, fg, np.float64)


def _assemble(po, p, parts, head):
    """One caller state (pyoracle.State, the reference's wa / iwa layout for all n rows) out of the ranks'
    exports of THEIR rows.  parts[r] = (x, g, wa, iwa) of rank r; head = (f, task, csave, lsave, isave, dsave),
    identical on every rank.  The host matrices are replicated (rank 0's are taken).  Index -- which the next
    freev reads to tell who enters and who leaves (:2012-2040) -- is put together from the ranks' local lists
    (export_state: free rows ascending from the front, the others from the back, Indx2(1) = the local number
    of free rows); the enter / leave halves of Indx2 are dead between calls and stay empty."""
    n, m = p.n, p.m
    off = po.wa_offsets(n, m)
    wa = np.zeros(po.wa_len(n, m))
    iwa = np.zeros(3 * n, np.int32)
    row0 = 0
    free_rows, act_rows = [], []
    for r, (x, g, wl, il) in enumerate(parts):
        nl = x.size
        ol = po.wa_offsets(nl, m)
        for name in ("ws", "wy"):
            a = wl[ol[name][0]:ol[name][0] + m * nl].reshape(m, nl)
            wa[off[name][0]:off[name][0] + m * n].reshape(m, n)[:, row0:row0 + nl] = a
        for name in ("z", "r", "d", "t", "xp"):
            wa[off[name][0] + row0:off[name][0] + row0 + nl] = wl[ol[name][0]:ol[name][0] + nl]
        if r == 0:
            for name in ("sy", "ss", "wt", "wn", "snd", "wa8m"):
                wa[off[name][0]:off[name][0] + off[name][1]] = wl[ol[name][0]:ol[name][0] + ol[name][1]]
        iwa[n + row0:n + row0 + nl] = il[nl:2 * nl]
        if il[:nl].any():                     # a freev has run
            nf = int(il[2 * nl])
            free_rows.append(il[:nf] + row0)
            act_rows.append(il[nl - 1:nf - 1 if nf > 0 else None:-1] + row0)
        row0 += nl
    if free_rows:
        fr, ac = np.concatenate(free_rows), np.concatenate(act_rows)
        assert fr.size + ac.size == n
        iwa[:fr.size] = fr
        iwa[n - 1:fr.size - 1 if fr.size > 0 else None:-1] = ac
    f, task, csave, lsave, isave, dsave = head
    return po.State(n, m, np.concatenate([q[0] for q in parts]), np.concatenate([q[1] for q in parts]),
                    np.array([f]), wa, iwa, task.copy(), csave.copy(), lsave.copy(), isave.copy(), dsave.copy())


def run_fuzz(rank, world, port, first, count, iters, out_path, family=None, scale=1, comm="gloo"):
    """`count` random problems, rows cut over `world` ranks sharing cuda:0 (gloo host reducers); the
    objective is evaluated per shard on the host and summed over the ranks.  Rank 0 follows the oracle's
    trajectory call by call; at the FIRST call that differs every rank hands over the state it exported
    before and after that call, rank 0 puts the n rows together and ONE oracle call from the sharded run's
    own previous state must reproduce the sharded run's call (tests/test_gpu_fuzz.py: _explain_divergence).
    family / scale: a problem of tests/test_gpu_fuzz.py's FAMILIES with its n multiplied (the objective is then
    evaluated on the gathered x by every rank); comm = "fakerccl": through the library's communicator code path."""
    import torch
    import torch.distributed as dist
    import lbfgsb_amd
    from oracle import pyoracle as po
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_fuzz as tf
    from test_gpu_fuzz import _explain_divergence

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank,
                            world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    results = {}

    def row(t, isave, f):
        return [t[:12], int(isave[29]), int(isave[33]), int(isave[32]), int(isave[37]), float(f)]
    for seed in range(first, first + count):
        if family:
            gen = (lambda q, t_: tf.make(q, t_, 2000, 1, 9)) if family == "make" else tf.FAMILIES[family]
            p = tf.scaled_up(gen, scale)(po, seed)
        else:
            p = fuzz_problem(po, seed)
        ro = []
        if rank == 0:
            po.run(po.Engine("oracle"), p, max_iter=iters, snapshot=lambda k, s: ro.append(row(s.task_s, s.isave, s.f[0])))
        row0, n_loc = lbfgsb_amd.block_partition(p.n, world, rank)
        sol = lbfgsb_amd.DeviceSolver(n_loc, p.m, n_global=p.n, row0=row0, device=0)
        if comm == "fakerccl":
            ids = [lbfgsb_amd.DeviceSolver.rccl_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, 0)
            sol.init_rccl(ids[0], rank, world)
            assert "libfake_rccl" in open("/proc/self/maps").read()
        else:
            lbfgsb_amd.attach_host_group(sol, rank, world)
        sl = slice(row0, row0 + n_loc)
        x = torch.from_numpy(p.x0[sl].copy()).to(dev)
        g = torch.zeros_like(x)
        l = torch.from_numpy(p.l[sl].copy()).to(dev)
        u = torch.from_numpy(p.u[sl].copy()).to(dev)
        nbd = torch.from_numpy(p.nbd[sl].astype(np.int32)).to(dev)
        rows, prev, split, verdict = [], None, None, None
        for k in range(100000):
            t = sol.setulb(x, l, u, nbd, g, 0.0, 0.0)
            rows.append(row(t, sol.isave, sol.f[0]))
            flag = torch.zeros(1, dtype=torch.int32)
            cur = None
            if split is None:
                torch.cuda.synchronize()
                wa, iwa = sol.export_state()
                cur = [x.cpu().numpy(), g.cpu().numpy(), wa, iwa]
                head = (float(sol.f[0]), sol.task.copy(), sol.csave.copy(), sol.lsave.copy(), sol.isave.copy(),
                        sol.dsave.copy())
                if rank == 0:
                    same = (k < len(ro) and ro[k][:5] == rows[k][:5]
                            and abs(ro[k][5] - rows[k][5]) <= 1e-8 * max(1.0, abs(ro[k][5])))
                    flag[0] = 0 if same else 1
                dist.broadcast(flag, 0)
                if int(flag[0]):
                    split = k
                    both = [None] * world
                    dist.gather_object((prev, (cur, head)), both if rank == 0 else None, 0)
                    if rank == 0:
                        try:
                            assert prev is not None, "diverged at the very first call"
                            sp = _assemble(po, p, [b[0][0] for b in both], both[0][0][1])
                            sc = _assemble(po, p, [b[1][0] for b in both], both[0][1][1])
                            _explain_divergence(po, p, sp, sc)
                            verdict = "reproduced"
                        except AssertionError as e:
                            verdict = "call %d (%s vs %s) NOT reproduced: %s" % (k, ro[k:k + 1], rows[k], str(e)[:300])
            if t.startswith("FG"):
                xh = x.cpu().numpy()
                if family:   # any objective: every rank evaluates it on the gathered x and keeps its rows
                    xs_all = [None] * world
                    dist.all_gather_object(xs_all, xh)
                    xf = np.concatenate(xs_all)
                    gf = np.empty_like(xf)
                    ft = torch.tensor([p.fg(xf, gf)], dtype=torch.float64)
                    gh = gf[sl].copy()
                else:
                    gh = np.empty_like(xh)
                    ft = torch.tensor([p.fg(xh, gh, row0, row0 + n_loc)], dtype=torch.float64)
                    dist.all_reduce(ft)
                g.copy_(torch.from_numpy(gh))
                sol.f[0] = float(ft[0])
                if cur is not None:   # the state the next call starts from: f, g as evaluated at the run's own x
                    cur[1] = gh.copy()
                    head = (float(ft[0]),) + head[1:]
            elif t.startswith("NEW_X"):
                if sol.isave[29] >= iters:
                    break
            else:
                break
            prev = (cur, head) if cur is not None else None
        results[str(seed)] = {"rows": [r_[1:] for r_ in rows if r_[0].startswith("NEW_X")], "task": sol.task_s,
                              "calls": len(rows), "oracle_calls": len(ro), "split": split, "verdict": verdict}
        sol.close()
    if rank == 0:
        with open(out_path, "w") as fh:
            json.dump(results, fh)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    a = sys.argv
    if a[4].startswith("fuzz"):   # rank world port "fuzz[:family:scale:comm]" first count iters - out
        opt = a[4].split(":")
        kw = dict(family=opt[1], scale=int(opt[2]), comm=opt[3]) if len(opt) == 4 else {}
        run_fuzz(int(a[1]), int(a[2]), int(a[3]), int(a[5]), int(a[6]), int(a[7]), a[9], **kw)
    else:
        run(int(a[1]), int(a[2]), int(a[3]), a[4], int(a[5]), int(a[6]), int(a[7]), a[8], a[9])

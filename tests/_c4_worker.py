"""Worker for BASELINE.json configs[3] on ONE GPU: the separable bounded quadratic at n = 1e8, m = 10,
fp64, cut into `world` contiguous row blocks -- `world` RANKS AS HOST THREADS of this one process
(one context per thread, every context on cuda:0), through the library's communicator code path
(ncclCommInitRank, ncclAllGather on each context's stream) with tests/fake_rccl.cpp standing in for
librccl (LBFGSB_RCCL_LIBRARY; real RCCL refuses ranks that share a GPU).  Threads, not processes: a
GPU box admits few processes on its card, and the C ABI's contract is one host thread per context.

    python tests/_c4_worker.py WORLD N M ITERS OUT.json
"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    world, n, m, iters, out_path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    import torch
    import lbfgsb_amd

    assert "fake_rccl" in os.environ.get("LBFGSB_RCCL_LIBRARY", ""), "this worker shares one GPU between the ranks"
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    uid = lbfgsb_amd.DeviceSolver.rccl_unique_id()   # (loads the stand-in library, once, on this thread)
    assert "libfake_rccl" in open("/proc/self/maps").read()
    results = [None] * world
    errors = []
    gate = threading.Barrier(world)

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            row0, n_loc = lbfgsb_amd.block_partition(n, world, rank)
            sol = lbfgsb_amd.DeviceSolver(n_loc, m, n_global=n, row0=row0, device=0, same_stream_objective=True)
            sol.init_rccl(uid, rank, world)
            x = torch.zeros(n_loc, dtype=torch.float64, device=dev)
            g = torch.zeros_like(x)
            l, u = torch.full_like(x, -1.0), torch.full_like(x, 1.0)
            nbd = torch.full((n_loc,), 2, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            gate.wait()
            rows, t_first, t0 = [], None, time.perf_counter()
            st_first = None
            while True:
                t = sol.setulb(x, l, u, nbd, g, 0.0, 0.0)
                if t.startswith("FG"):
                    sol.objective(0, x, g, deferred=True)
                elif t.startswith("NEW_X"):
                    rows.append([int(sol.isave[29]), int(sol.isave[33]), int(sol.isave[32]), int(sol.isave[37]),
                                 float(sol.f[0]), float(sol.dsave[12]), int(sol.isave[27])])
                    if t_first is None:
                        sol.sync()
                        t_first = time.perf_counter() - t0
                        st_first = sol.stats()
                    if sol.isave[29] >= iters:
                        break
                else:
                    break
            sol.sync()
            t_all = time.perf_counter() - t0
            st = sol.stats()
            results[rank] = {"rank": rank, "row0": row0, "n_loc": n_loc, "rows": rows, "task": sol.task_s,
                             "first_iteration_s": t_first, "total_s": t_all, "stats": st, "stats_after_first": st_first,
                             "tie_splits": sol.tie_splits(), "path_counts": list(sol.path_counts()),
                             "x_head": x[:4].cpu().numpy().tolist()}
            gate.wait()
            sol.close()
        except BaseException as e:   # noqa: BLE001
            errors.append((rank, repr(e)))
            try:
                gate.abort()
            except Exception:
                pass
    threads = [threading.Thread(target=rank_main, args=(r,), name="rank%d" % r) for r in range(world)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    with open(out_path, "w") as fh:
        json.dump({"world": world, "n": n, "m": m, "iters": iters, "errors": errors, "ranks": results}, fh)
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main())

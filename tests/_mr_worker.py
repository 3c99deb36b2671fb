"""Worker for the multi-rank tests: `world` processes, one shard of the rows each.
mode 'gloo': ranks share cuda:0, reductions through a gloo host group (host callbacks).
mode 'rccl1': world must be 1; an RCCL communicator of one rank is attached so that the
ncclAllReduce / ncclAllGather code path runs on a single GPU."""
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
    sol = lbfgsb_amd.DeviceSolver(n_loc, m, n_global=n, row0=row0, device=0,
                                  parallel_gcp=(variant == "pgcp"))
    if mode == "gloo":
        lbfgsb_amd.attach_host_group(sol, rank, world)
    elif mode == "rccl1":
        assert world == 1
        lbfgsb_amd.attach_rccl(sol, 0, 1, dev)
    if variant == "rosen":
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
    for _ in range(100000):
        t = sol.setulb(x, l, u, nbd, g, 0.0, 0.0)
        if t.startswith("FG"):
            sol.f[0] = sol.objective(kind, x, g)   # global f (reduced over ranks)
        elif t.startswith("NEW_X"):
            rows.append([int(sol.isave[29]), int(sol.isave[33]), int(sol.isave[32]),
                         int(sol.isave[37]), float(sol.f[0]), float(sol.dsave[12])])
            if sol.isave[29] >= iters:
                break
        else:
            break
    torch.cuda.synchronize()
    xs = [None] * world
    dist.all_gather_object(xs, x.cpu().numpy())
    if rank == 0:
        with open(out_path, "w") as fh:
            json.dump({"rows": rows, "task": sol.task_s, "x": np.concatenate(xs).tolist(),
                       "stats": sol.stats()}, fh)
    sol.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    a = sys.argv
    run(int(a[1]), int(a[2]), int(a[3]), a[4], int(a[5]), int(a[6]), int(a[7]), a[8], a[9])

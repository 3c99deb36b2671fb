"""Row sharding and reducers for the multi-GPU path.

One process per GPU.  The n rows are cut into contiguous blocks; every n-length
reduction of the L-BFGS-B iteration leaves the kernels as <= 4m+5 fp64 partials per
rank and is completed either by RCCL on the solver's stream (`attach_rccl`, the bench
path) or by a host-side `torch.distributed` group such as gloo (`attach_host_group`,
used when several ranks share one GPU and in the CPU tests).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def block_partition(n: int, world: int, rank: int) -> Tuple[int, int]:
    """(row0, n_local) of `rank`: contiguous blocks, the first n % world ranks get one more row."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad world/rank")
    base, rem = divmod(int(n), world)
    n_local = base + (1 if rank < rem else 0)
    row0 = rank * base + min(rank, rem)
    return row0, n_local


def make_group_reducers(group=None):
    """(allreduce, allgather) closures over a torch.distributed process group (gloo or any
    backend that takes CPU tensors), in the calling convention of
    DeviceSolver.init_host_reducer: allreduce(view, nsum, nmin, nmax) reduces the fp64
    numpy view in place (sums | mins | maxes); allgather(local_bytes) -> rank-major bytes."""
    import torch
    import torch.distributed as dist

    def allreduce(view: np.ndarray, nsum: int, nmin: int, nmax: int) -> None:
        t = torch.from_numpy(view)      # shares memory with the C buffer
        if nsum:
            dist.all_reduce(t[:nsum], op=dist.ReduceOp.SUM, group=group)
        if nmin:
            dist.all_reduce(t[nsum:nsum + nmin], op=dist.ReduceOp.MIN, group=group)
        if nmax:
            dist.all_reduce(t[nsum + nmin:nsum + nmin + nmax], op=dist.ReduceOp.MAX, group=group)

    def allgather(local: np.ndarray) -> np.ndarray:
        world = dist.get_world_size(group)
        src = torch.from_numpy(np.ascontiguousarray(local))
        out = [torch.empty_like(src) for _ in range(world)]
        dist.all_gather(out, src, group=group)
        return np.concatenate([o.numpy() for o in out])

    return allreduce, allgather


def attach_host_group(solver, rank: int, world: int, group=None) -> None:
    ar, ag = make_group_reducers(group)
    solver.init_host_reducer(ar, rank, world, allgather=ag)


def attach_rccl(solver, rank: int, world: int, device) -> None:
    """Create the RCCL communicator of the solver: rank 0 makes the ncclUniqueId, it is
    broadcast as 128 bytes through the default torch.distributed group."""
    import torch
    import torch.distributed as dist
    idt = torch.zeros(128, dtype=torch.uint8, device=device)
    if rank == 0:
        raw = type(solver).rccl_unique_id()
        idt.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
    if world > 1:
        dist.broadcast(idt, 0)
    solver.init_rccl(bytes(idt.cpu().numpy().tobytes()), rank, world)

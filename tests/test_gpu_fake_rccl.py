"""The stand-in for librccl that the multi-rank tests run on (tests/fake_rccl.cpp) must be able to FAIL
the way RCCL fails, or the 8-rank path's only safety net cannot see a regression:
  * a collective returns BEFORE its data have moved (asynchronous on the caller's stream);
  * ranks that pass different counts are caught: the result is poisoned, the next call is refused.
Two rank threads of this process, one GPU, the stand-in driven directly through ctypes."""
import ctypes as C
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def _lib():
    lib = C.CDLL(os.environ["LBFGSB_FAKE_RCCL_SO"])
    lib.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    lib.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    lib.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.ncclCommUserRank.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    return lib


NCCL_DOUBLE = 8   # ncclFloat64 (rccl.h)


def _two_ranks(body):
    """run body(rank, comm, lib, stream, torch) on two threads with a 2-rank communicator each"""
    import torch
    lib = _lib()
    uid = UniqueId()
    assert lib.ncclGetUniqueId(C.byref(uid)) == 0
    out, errs = [None, None], []

    def worker(rank):
        try:
            torch.cuda.set_device(0)
            comm = C.c_void_p()
            assert lib.ncclCommInitRank(C.byref(comm), 2, uid, rank) == 0
            n, r = C.c_int(), C.c_int()
            assert lib.ncclCommCount(comm, C.byref(n)) == 0 and lib.ncclCommUserRank(comm, C.byref(r)) == 0
            assert (n.value, r.value) == (2, rank)
            stream = torch.cuda.Stream()
            out[rank] = body(rank, comm, lib, stream, torch)
            stream.synchronize()
            out[rank] = (out[rank], lib.ncclCommDestroy(comm))
        except BaseException as e:   # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not any(t.is_alive() for t in ts), "a rank thread hangs"
    if errs:
        raise errs[0]
    return out


def scenario_async():
    def body(rank, comm, lib, stream, torch):
        k = 1000
        send = torch.full((k,), float(rank + 1), dtype=torch.float64, device="cuda")
        recv = torch.full((2 * k,), -7.0, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            # plenty of work in front of the collective on the same stream (~tens of ms) ...
            a = torch.ones((4096, 4096), device="cuda")
            for _ in range(40):
                a = (a @ a) * 1e-4
            rc = lib.ncclAllGather(send.data_ptr(), recv.data_ptr(), k, NCCL_DOUBLE, comm,
                                   C.c_void_p(stream.cuda_stream))
            # ... so when the call RETURNS the stream has not reached it: nothing has been written yet
            returned_before_done = not stream.query()
        assert rc == 0
        stream.synchronize()
        got = recv.cpu().numpy()
        assert np.all(got[:k] == 1.0) and np.all(got[k:] == 2.0)
        return bool(returned_before_done)
    return _two_ranks(body)


def scenario_mismatch():
    def body(rank, comm, lib, stream, torch):
        k = 64 + rank    # rank 0 passes 64 elements, rank 1 passes 65
        send = torch.full((65,), float(rank + 1), dtype=torch.float64, device="cuda")
        recv = torch.zeros(130, dtype=torch.float64, device="cuda")
        rc1 = lib.ncclAllGather(send.data_ptr(), recv.data_ptr(), k, NCCL_DOUBLE, comm,
                                C.c_void_p(stream.cuda_stream))
        stream.synchronize()
        poisoned = bool(np.all(np.isnan(recv.cpu().numpy()[:2 * k])))
        rc2 = lib.ncclAllGather(send.data_ptr(), recv.data_ptr(), 64, NCCL_DOUBLE, comm,
                                C.c_void_p(stream.cuda_stream))
        return rc1, poisoned, rc2
    return _two_ranks(body)


def _child(name):
    """the two rank threads live in a child process: each rank's stream needs a hardware queue of its own
    (GPU_MAX_HW_QUEUES, read when the runtime starts) -- a rank's stream-side wait must not sit in front of
    the other rank's copies"""
    from test_gpu_multirank import _fake_rccl
    env = dict(os.environ, LBFGSB_FAKE_RCCL_SO=_fake_rccl(), GPU_MAX_HW_QUEUES="8")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), name], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1]), r.stderr


def test_all_gather_is_asynchronous_and_correct():
    out, _ = _child("async")
    assert out[0][0] and out[1][0], "ncclAllGather blocked until its data had moved"
    assert out[0][1] == 0 and out[1][1] == 0


def test_unequal_counts_are_caught():
    out, err = _child("mismatch")
    for (rc1, poisoned, rc2), rc_destroy in out:
        assert rc1 == 0             # (asynchronous: the call itself cannot know yet)
        assert poisoned             # the result of the mismatched collective is NaN, not plausible numbers
        assert rc2 != 0             # ... and the communicator refuses to go on
        assert rc_destroy != 0
    assert "real RCCL would hang or corrupt here" in err


if __name__ == "__main__":
    sys.path.insert(0, HERE)
    res = {"async": scenario_async, "mismatch": scenario_mismatch}[sys.argv[1]]()
    print(json.dumps(res))

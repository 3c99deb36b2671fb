"""ctypes bindings of include/lbfgsb_hip.h and include/lbfgsb_hip_debug.h (one prototype per declared symbol)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# every symbol include/lbfgsb_hip.h declares: name -> (restype, argtypes)
_vp, _i32p, _dp, _cp = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_char_p
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)
FG_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_void_p, C.c_void_p)
PROTOTYPES = {
    "lbfgsb_hip_create": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int,
                                    _vp, C.POINTER(_vp)]),
    "lbfgsb_hip_destroy": (None, [_vp]),
    "lbfgsb_hip_last_error": (C.c_char_p, []),
    "lbfgsb_hip_rccl_unique_id": (C.c_int, [_vp]),
    "lbfgsb_hip_comm_init_rccl": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "lbfgsb_hip_comm_init_host": (C.c_int, [_vp, ALLREDUCE_FN, ALLGATHER_FN, _vp, C.c_int, C.c_int]),
    "lbfgsb_hip_setulb_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double,
                                        _vp, C.c_int, _vp, _vp, _vp, _vp]),
    "lbfgsb_hip_setulb_dev_pp": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double,
                                           C.c_double, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "lbfgsb_hip_setulb_host": (C.c_int, [C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp,
                                         C.c_double, C.c_double, _vp, _vp, _vp, C.c_int32, _vp,
                                         _vp, _vp, _vp, _cp, C.c_int32, C.c_int32]),
    "lbfgsb_hip_setulb_host_ik": (C.c_int, [C.c_int64, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp,
                                            C.c_double, C.c_double, _vp, _vp, _vp, C.c_int64, _vp,
                                            _vp, _vp, _vp, _cp, C.c_int32, C.c_int32, C.c_int32]),
    "lbfgsb_hip_release_host_ik": (C.c_int, [_vp, C.c_int32]),
    "lbfgsb_hip_pass_clock": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "lbfgsb_hip_get_stream": (C.c_void_p, [_vp]),
    "lbfgsb_hip_wait_stream": (C.c_int, [_vp, _vp]),
    "lbfgsb_hip_release_host": (C.c_int, [_vp]),
    "lbfgsb_hip_host_pinning": (C.c_int, [C.c_int]),
    "lbfgsb_hip_return_event": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "lbfgsb_hip_f_device": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "lbfgsb_hip_tie_splits": (C.c_int, [_vp, _vp]),
    "lbfgsb_hip_defer_stats": (C.c_int, [_vp, _vp, _vp]),
    "lbfgsb_hip_host_gap": (C.c_int, [_vp, _vp, _vp]),
    "lbfgsb_hip_collective_time": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "lbfgsb_hip_compact_stats": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "lbfgsb_hip_host_segments": (C.c_int, [_vp, _vp]),
    "lbfgsb_hip_comm_info": (C.c_int, [_vp, _vp, _vp, _vp]),
    "lbfgsb_hip_path_counts": (C.c_int, [_vp, _vp, _vp, _vp]),
    "lbfgsb_hip_minimize": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_int,
                                      C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "lbfgsb_hip_export_state": (C.c_int, [_vp, _vp, _vp]),
    "lbfgsb_hip_import_state": (C.c_int, [_vp, _vp, _vp, _vp]),
    "lbfgsb_hip_projgr": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lbfgsb_hip_wtv": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp]),
    "lbfgsb_hip_set_w": (C.c_int, [_vp, _vp, _vp]),
    "lbfgsb_hip_set_iwhere": (C.c_int, [_vp, _vp]),
    "lbfgsb_hip_formk_gram": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "lbfgsb_hip_wtv_launch_only": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "lbfgsb_hip_wtv_time": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "lbfgsb_hip_kernel_time": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "lbfgsb_hip_sync": (C.c_int, [_vp]),
    "lbfgsb_hip_objective": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "lbfgsb_hip_stats": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "lbfgsb_hip_set_option": (C.c_int, [_vp, _cp, C.c_double]),
    "lbfgsb_hip_comm_stats": (C.c_int, [_vp, _vp, _vp]),
    "lbfgsb_hip_uniform_bounds": (C.c_int, [_vp, _vp]),
    "lbfgsb_hip_freev_skipped": (C.c_int, [_vp, _vp]),
    "lbfgsb_hip_skip_stats": (C.c_int, [_vp, _vp]),
    "lbfgsb_hip_refresh_count": (C.c_int, [_vp, _vp]),
    "lbfgsb_hip_vec_sub": (C.c_int, [_vp, _vp, _vp, _vp]),
    "lbfgsb_hip_vec_scale": (C.c_int, [_vp, C.c_double, _vp]),
    "lbfgsb_hip_dot": (C.c_int, [_vp, _vp, _vp, _vp]),
    # routine doors
    "lbfgsb_hip_active": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "lbfgsb_hip_errclb": (C.c_int, [_vp, _vp, _vp, _vp, C.c_double, _vp, _vp, _vp]),
    "lbfgsb_hip_cauchy": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_int, C.c_int, C.c_double,
                                    _vp, _vp, _vp]),
    "lbfgsb_hip_freev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "lbfgsb_hip_formk": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, _vp]),
    "lbfgsb_hip_cmprlb": (C.c_int, [_vp, _vp, _vp, C.c_double, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "lbfgsb_hip_subsm": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_int, C.c_int, _vp, _vp,
                                   _vp]),
    "lbfgsb_hip_lnsrlb": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lbfgsb_hip_matupd": (C.c_int, [_vp, _vp, C.c_double, C.c_double, C.c_double, _vp, _vp]),
}

E_NOGPU, E_ARG, E_ALLOC, E_COMM, E_STATE = -100, -101, -102, -103, -104   # status codes of include/lbfgsb_hip.h
F_REAL32 = 1
F_MIRROR_INDEX = 2
F_NO_RETURN_SYNC = 4
F_PARALLEL_GCP = 8
F_EXACT_TIES = 16      # accepted and ignored: the default since round 3
F_INDEX_TIES = 32      # opt-out: equal breakpoints in variable order, no heap-order replay
F_DEFER_LNSRCH = 64    # the line-search set-up's sums travel with the next call's fetch (same-stream objective)


class LbfgsbError(RuntimeError):
    pass


def lib_path() -> str:
    # LBFGSB_HIP_LIBRARY: another build of the same ABI (A/B timing of two builds on one box)
    return os.environ.get("LBFGSB_HIP_LIBRARY") or os.path.join(HERE, "liblbfgsb_hip.so")


def build_library() -> str:
    """hipcc --offload-arch=gfx950 ... (cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "csrc")])
    return lib_path()


def load_library():
    """Load the HIP library.  torch is imported first so that both share one HIP runtime."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise LbfgsbError("%s is missing: run lbfgsb_amd.build_library() (hipcc, gfx950). "
                          "There is no CPU fallback." % path)
    try:
        import torch  # noqa: F401  (loads libamdhip64 that this library then binds to)
    except Exception:
        pass
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(rc: int):
    if rc != 0:
        msg = load_library().lbfgsb_hip_last_error()
        raise LbfgsbError("lbfgsb_hip error %d: %s" % (rc, (msg or b"").decode()))

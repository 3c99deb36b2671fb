"""lbfgsb_amd -- MI355X-native L-BFGS-B inner iteration behind the reference's
reverse-communication `setulb` API (jacobwilliams/lbfgsb, src/lbfgsb.f90:88).

The compute path is the hand-written HIP library `liblbfgsb_hip.so` (C ABI in
include/lbfgsb_hip.h).  There is no CPU fallback: importing this package works
anywhere, but every call that computes raises `LbfgsbError` when the library or
a gfx950 device is missing.
"""
from .capi import LbfgsbError, lib_path, load_library, build_library  # noqa: F401
from .solver import DeviceSolver, setulb, wa_length, TASK_LEN  # noqa: F401
from .distributed import block_partition, attach_rccl, attach_host_group  # noqa: F401

__all__ = ["LbfgsbError", "DeviceSolver", "setulb", "wa_length", "load_library",
           "build_library", "lib_path", "TASK_LEN"]

"""pyoracle -- ctypes access to the CPU oracle (TEST INFRASTRUCTURE ONLY).

Loads oracle/liblbfgsb_oracle.so (our plain-C restatement, lbfgsb_oracle.c) and,
when present, oracle/_ref/liblbfgsb_ref*.so (the real reference built from
/root/reference by oracle/Makefile).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg import this module; the product package
(lbfgsb_amd/) never does.

Nothing here reads /root/reference at run time.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import Callable, List, Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

_c_int_p = C.POINTER(C.c_int)


def build(ref: bool = True) -> None:
    """Compile the oracle (and the reference, if its sources are here)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if ref:
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Engine:
    """A setulb-compatible CPU engine: the oracle ('oracle') or the real
    reference ('ref', 'ref_i8', 'ref_r32', 'ref_r32_i8')."""

    def __init__(self, kind: str = "oracle"):
        self.kind = kind
        self.real = np.float32 if "r32" in kind else np.float64
        self.int = np.int64 if kind.endswith("i8") else np.int32
        creal = C.c_float if self.real == np.float32 else C.c_double
        cint = C.c_int64 if self.int == np.int64 else C.c_int
        if kind.startswith("oracle"):
            path = os.path.join(HERE, "liblbfgsb_oracle%s.so" % ("_r32" if kind.endswith("r32") else ""))
            sym = "lbo_setulb"
        else:
            path = os.path.join(HERE, "_ref", "liblbfgsb_%s.so" % kind)
            sym = "ref_setulb"
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.path = path
        self.lib = C.CDLL(path)
        self.fn = getattr(self.lib, sym)
        self.fn.restype = None
        self.fn.argtypes = [cint, cint, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                            C.c_void_p, C.c_void_p, creal, creal, C.c_void_p, C.c_void_p,
                            C.c_void_p, cint, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        if kind.startswith("oracle"):
            L = self.lib
            L.lbo_quadratic_fg.restype = creal
            L.lbo_quadratic_fg.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
            L.lbo_rosenbrock_fg.restype = creal
            L.lbo_rosenbrock_fg.argtypes = [C.c_int64, C.c_void_p, C.c_void_p]

    def available(kind: str) -> bool:  # type: ignore[misc]
        try:
            Engine(kind)
            return True
        except (FileNotFoundError, OSError):
            return False

    available = staticmethod(available)  # type: ignore[assignment]


class Routines:
    """The oracle's routines one by one (lbo_active ... lbo_matupd, each following the reference routine
    its header in lbfgsb_oracle.c cites): the CPU twins of the library's routine doors
    (tests/test_gpu_routines.py).  Arrays are numpy (real / int32), scalars Python numbers; in/out
    scalars are one-element arrays."""

    def __init__(self, real=np.float64):
        self.real = real
        eng = Engine("oracle_r32" if real == np.float32 else "oracle")
        L = self.lib = eng.lib
        R = C.c_float if real == np.float32 else C.c_double
        I, P = C.c_int, C.c_void_p
        sig = {
            "lbo_active": [I, P, P, P, P, P, P, P, P, P],
            "lbo_errclb": [I, I, R, P, P, P, P, P, P],
            "lbo_cauchy": [I, P, P, P, P, P, P, P, P, P, P, I, P, P, P, P, R, I, I, P, P, P, P, P, R, P, R],
            "lbo_cmprlb": [I, I, P, P, P, P, P, P, P, P, P, P, R, I, I, I, I, P],
            "lbo_freev": [I, P, P, P, P, P, P, P, I, I, I],
            "lbo_formk": [I, I, P, I, I, P, I, I, P, P, I, P, P, P, R, I, I, P],
            "lbo_formt": [I, P, P, P, I, R, P],
            "lbo_matupd": [I, I, P, P, P, P, P, P, P, I, P, P, P, R, R, R, R],
            "lbo_subsm": [I, I, I, P, P, P, P, P, P, P, P, P, R, P, P, I, I, P, P, P, P],
            "lbo_lnsrlb": [I, P, P, P, P, R, P, P, P, P, P, P, P, P, P, P, P, P, P, I, P, P, P, P, P, I, I, P, P,
                           P],
        }
        for name, args in sig.items():
            fn = getattr(L, name)
            fn.restype = None
            fn.argtypes = args
        L.lbo_ddot.restype = R
        L.lbo_ddot.argtypes = [C.c_int64, P, P]
        L.lbo_projgr.restype = R
        L.lbo_projgr.argtypes = [I, P, P, P, P, P]

    def __getattr__(self, name):
        fn = getattr(self.lib, "lbo_" + name)

        def call(*args):
            return fn(*[_ptr(a) if isinstance(a, np.ndarray) else a for a in args])
        return call


# --------------------------------------------------------------------------
# objectives (computed by the oracle's C code so that every engine is fed
# bit-identical f, g)
# --------------------------------------------------------------------------
_OBJ_LIB = {}


def _objlib(real):
    key = "r32" if real == np.float32 else "r64"
    if key not in _OBJ_LIB:
        _OBJ_LIB[key] = Engine("oracle_r32" if real == np.float32 else "oracle")
    return _OBJ_LIB[key].lib


def quadratic_fg(x: np.ndarray, g: np.ndarray, i0: int = 0) -> float:
    """SURVEY.md 8d separable bounded quadratic; i0 = global 0-based offset."""
    return float(_objlib(x.dtype.type).lbo_quadratic_fg(x.size, i0, _ptr(x), _ptr(g)))


def rosenbrock_fg(x: np.ndarray, g: np.ndarray) -> float:
    """reference test/driver1.f90:274-289."""
    return float(_objlib(x.dtype.type).lbo_rosenbrock_fg(x.size, _ptr(x), _ptr(g)))


@dataclass
class Problem:
    name: str
    n: int
    m: int
    x0: np.ndarray
    l: np.ndarray
    u: np.ndarray
    nbd: np.ndarray
    factr: float
    pgtol: float
    fg: Callable[[np.ndarray, np.ndarray], float]
    real: type = np.float64


def problem_rosenbrock(n=25, m=5, factr=1e7, pgtol=1e-5, real=np.float64) -> Problem:
    """driver1/2/3 problem (test/driver1.f90:233-251)."""
    l = np.empty(n, real)
    u = np.full(n, 100.0, real)
    l[0::2] = 1.0
    l[1::2] = -100.0
    return Problem("rosenbrock", n, m, np.full(n, 3.0, real), l, u, np.full(n, 2, np.int32),
                   factr, pgtol, rosenbrock_fg, real)


def problem_quadratic(n=1000, m=10, mixed_nbd=False, real=np.float64) -> Problem:
    """BASELINE.md section 3 problem; mixed_nbd: nbd_i = mod(i,4) (1-based i)."""
    nbd = np.full(n, 2, np.int32)
    if mixed_nbd:
        nbd = (np.arange(1, n + 1) % 4).astype(np.int32)
    return Problem("quadratic_mixed" if mixed_nbd else "quadratic", n, m, np.zeros(n, real),
                   np.full(n, -1.0, real), np.full(n, 1.0, real), nbd, 0.0, 0.0,
                   lambda x, g: quadratic_fg(x, g, 0), real)


# --------------------------------------------------------------------------
# reverse-communication state + driver loop
# --------------------------------------------------------------------------
def pad60(s: str) -> np.ndarray:
    b = s.encode()[:60]
    return np.frombuffer(b + b" " * (60 - len(b)), dtype=np.uint8).copy()


def task_str(a: np.ndarray) -> str:
    return bytes(a.tobytes()).decode("ascii", "replace").rstrip()


def wa_len(n: int, m: int) -> int:
    return 2 * m * n + 5 * n + 11 * m * m + 8 * m


def wa_offsets(n: int, m: int) -> dict:
    """0-based slot offsets, reference src/lbfgsb.f90:250-265."""
    names = ["ws", "wy", "sy", "ss", "wt", "wn", "snd", "z", "r", "d", "t", "xp", "wa8m"]
    sizes = [m * n, m * n, m * m, m * m, m * m, 4 * m * m, 4 * m * m, n, n, n, n, n, 8 * m]
    off, o = {}, 0
    for k, s in zip(names, sizes):
        off[k] = (o, s)
        o += s
    return off


@dataclass
class State:
    """Every caller-owned array of the reverse-communication API."""
    n: int
    m: int
    x: np.ndarray
    g: np.ndarray
    f: np.ndarray
    wa: np.ndarray
    iwa: np.ndarray
    task: np.ndarray
    csave: np.ndarray
    lsave: np.ndarray
    isave: np.ndarray
    dsave: np.ndarray

    @staticmethod
    def fresh(p: Problem, int_dtype=np.int32) -> "State":
        return State(p.n, p.m, p.x0.copy(), np.zeros(p.n, p.real), np.zeros(1, p.real),
                     np.zeros(wa_len(p.n, p.m), p.real), np.zeros(3 * p.n, int_dtype),
                     pad60("START"), pad60(""), np.zeros(4, int_dtype),
                     np.zeros(44, int_dtype), np.zeros(29, p.real))

    def copy(self) -> "State":
        return State(self.n, self.m, *[a.copy() for a in (self.x, self.g, self.f, self.wa,
                     self.iwa, self.task, self.csave, self.lsave, self.isave, self.dsave)])

    @property
    def task_s(self) -> str:
        return task_str(self.task)


def call(engine: Engine, p: Problem, s: State, iprint: int = -1) -> None:
    nbd = p.nbd.astype(engine.int, copy=False)
    nbd = np.ascontiguousarray(nbd)
    engine.fn(p.n, p.m, _ptr(s.x), _ptr(p.l), _ptr(p.u), _ptr(nbd), _ptr(s.f), _ptr(s.g),
              p.factr, p.pgtol, _ptr(s.wa), _ptr(s.iwa), _ptr(s.task), iprint, _ptr(s.csave),
              _ptr(s.lsave), _ptr(s.isave), _ptr(s.dsave))


def run(engine: Engine, p: Problem, max_calls: int = 10**9, max_iter: int = 10**9,
        snapshot: Optional[Callable[[int, State], None]] = None,
        on_new_x: Optional[Callable[[State], Optional[str]]] = None) -> State:
    """Drive the reverse-communication loop like test/driver1.f90:263-292.
    snapshot(k, state) is called after the k-th setulb return (k = 0, 1, ...).
    on_new_x may return a 'STOP...' string (driver2/3 style user stop)."""
    s = State.fresh(p, engine.int)
    k = 0
    while k < max_calls:
        call(engine, p, s)
        if snapshot is not None:
            snapshot(k, s)
        k += 1
        t = s.task_s
        if t.startswith("FG"):
            s.f[0] = p.fg(s.x, s.g)
        elif t.startswith("NEW_X"):
            if s.isave[29] >= max_iter:
                break
            if on_new_x is not None:
                stop = on_new_x(s)
                if stop:
                    s.task[:] = pad60(stop)
        else:
            break
    return s

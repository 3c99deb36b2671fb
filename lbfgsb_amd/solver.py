"""Host-side mirror of the reference interface for the hot path.

`setulb(...)` has the reference's argument list (src/lbfgsb.f90:88-89) with numpy
arrays standing in for the Fortran arrays; it calls the C ABI's host-pointer form
exactly as the Fortran shim does.  `DeviceSolver` is the device-resident form used
by bench.py and the parity tests: x, l, u, nbd, g are torch tensors on the GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import capi
from .capi import LbfgsbError, check, load_library

TASK_LEN = 60


def wa_length(n: int, m: int) -> int:
    """reference src/lbfgsb.f90:146"""
    return 2 * m * n + 5 * n + 11 * m * m + 8 * m


def _p(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(int(a.data_ptr()))  # torch tensor


def pad60(s: str) -> np.ndarray:
    b = s.encode()[:TASK_LEN]
    return np.frombuffer(b + b" " * (TASK_LEN - len(b)), dtype=np.uint8).copy()


def task_str(a) -> str:
    return bytes(np.asarray(a).tobytes()).decode("ascii", "replace").rstrip()


def setulb(n, m, x, l, u, nbd, f, g, factr, pgtol, wa, iwa, task, iprint, csave, lsave, isave,
           dsave, iteration_file: Optional[str] = None, mirror: bool = False):
    """Reference `setulb` (src/lbfgsb.f90:88).  All arrays are numpy and are updated in
    place: x, g, wa (real kind), f (1 element), iwa/isave/lsave (int32), task/csave
    (60 uint8, blank padded), dsave (real kind).  mirror=True exports the complete
    reference layout of wa/iwa after every return (parity tests)."""
    lib = load_library()
    real_bytes = x.dtype.itemsize
    check(lib.lbfgsb_hip_setulb_host(
        n, m, _p(x), _p(l), _p(u), _p(nbd), _p(f), _p(g), float(factr), float(pgtol), _p(wa),
        _p(iwa), _p(task), int(iprint), _p(csave), _p(lsave), _p(isave), _p(dsave),
        iteration_file.encode() if iteration_file else None, real_bytes, 1 if mirror else 0))


class DeviceSolver:
    """One rank of the device-resident solver.  Tensors stay on the GPU; only the
    60-byte task, f and the isave/dsave scalars cross PCIe."""

    def __init__(self, n_local: int, m: int, n_global: Optional[int] = None, row0: int = 0,
                 real32: bool = False, mirror_index: bool = False, device: int = 0, stream=None,
                 same_stream_objective: bool = False, parallel_gcp: bool = False,
                 exact_ties: bool = True, index_ties: bool = False, options: Optional[dict] = None,
                 defer_lnsrch: bool = False, stream_ordered: bool = False):
        self.lib = load_library()
        # LBFGSB_F_DEFER_LNSRCH returns 'FG_LNSRCH' without waiting for the pass that writes the trial point
        # (it implies LBFGSB_F_NO_RETURN_SYNC): only a caller whose objective runs on the solver's OWN stream
        # may use it -- one that evaluates on torch's current stream would read x while it is being written
        # stream_ordered=True: the objective runs on torch's CURRENT stream and is ordered against the solver's stream
        # with events in both directions -- after every 'FG...' return torch's stream is made to wait for the trial
        # point (lbfgsb_hip_return_event), before every 'FG...' re-entry the solver's stream for g
        # (lbfgsb_hip_wait_stream): no host sync at an FG return either, so such a caller may defer as well.
        # f travels as a device scalar (set_f_device) or as a host float (sol.f[0] = float(...), which syncs torch).
        if defer_lnsrch and not (same_stream_objective or stream_ordered):
            raise ValueError("defer_lnsrch=True needs same_stream_objective=True or stream_ordered=True: the flag "
                             "skips the host sync at every FG_LNSRCH return, so f,g must be evaluated on the "
                             "solver's stream (DeviceSolver.objective / DeviceSolver.stream) or on a stream that is "
                             "ordered behind it with events (stream_ordered)")
        self.same_stream_objective = bool(same_stream_objective)
        self.stream_ordered = bool(stream_ordered) and not self.same_stream_objective
        self.n, self.m = int(n_local), int(m)
        self.n_global = int(n_global if n_global is not None else n_local)
        self.row0 = int(row0)
        self.real = np.float32 if real32 else np.float64
        flags = (capi.F_REAL32 if real32 else 0) | (capi.F_MIRROR_INDEX if mirror_index else 0)
        flags |= capi.F_NO_RETURN_SYNC if (same_stream_objective or stream_ordered) else 0
        flags |= capi.F_PARALLEL_GCP if parallel_gcp else 0  # opt-in, see include/lbfgsb_hip.h
        # a walk that ends inside a group of equal breakpoints is replayed in the reference's heap
        # order by default; index_ties=True (or exact_ties=False) opts out (include/lbfgsb_hip.h)
        flags |= capi.F_INDEX_TIES if (index_ties or not exact_ties) else 0
        # the caller evaluates f,g on the solver's stream and re-enters with nothing in between: the
        # line-search set-up's sums ride with the next call's fetch (include/lbfgsb_hip.h)
        flags |= capi.F_DEFER_LNSRCH if defer_lnsrch else 0
        h = C.c_void_p()
        sp = C.c_void_p(int(stream)) if stream else None
        check(self.lib.lbfgsb_hip_create(self.n, self.n_global, self.row0, self.m, flags, device,
                                         sp, C.byref(h)))
        self.h = h
        self.task = pad60("START")
        self.csave = pad60("")
        self.lsave = np.zeros(4, np.int32)
        self.isave = np.zeros(44, np.int32)
        self.dsave = np.zeros(29, np.float64)
        self.f = np.zeros(1, np.float64)
        self._cur = C.c_int32(0)
        self._cur_ref = C.byref(self._cur)
        self._keep = []
        for k, v in (options or {}).items():
            self.set_option(k, v)

    def set_option(self, name: str, value: float):
        """measurement / test switch of this context (lbfgsb_hip_set_option)"""
        if name == "defer_lnsrch" and float(value) != 0.0 and not (self.same_stream_objective or self.stream_ordered):
            raise ValueError("option defer_lnsrch needs a context created with same_stream_objective=True or "
                             "stream_ordered=True")
        check(self.lib.lbfgsb_hip_set_option(self.h, name.encode(), float(value)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.lbfgsb_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- multi-GPU ----
    def init_rccl(self, unique_id: bytes, rank: int, nranks: int):
        buf = C.create_string_buffer(unique_id, 128)
        check(self.lib.lbfgsb_hip_comm_init_rccl(self.h, buf, rank, nranks))

    @staticmethod
    def rccl_unique_id() -> bytes:
        lib = load_library()
        buf = C.create_string_buffer(128)
        check(lib.lbfgsb_hip_rccl_unique_id(buf))
        return buf.raw

    def init_host_reducer(self, allreduce, rank: int, nranks: int, allgather=None):
        """allreduce(np_view, nsum, nmin, nmax) reduces the fp64 view in place over ranks
        (sums | mins | maxes).  allgather(np_uint8_local) -> rank-major concatenation."""
        def _ar(user, buf, nsum, nmin, nmax):
            try:
                k = nsum + nmin + nmax
                view = np.ctypeslib.as_array(buf, shape=(k,))
                allreduce(view, nsum, nmin, nmax)
                return 0
            except Exception:  # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1

        def _ag(user, inp, out, nbytes):
            try:
                world = nranks
                src = np.ctypeslib.as_array(C.cast(inp, C.POINTER(C.c_uint8)), shape=(nbytes,))
                dst = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)),
                                            shape=(nbytes * world,))
                dst[:] = allgather(src)
                return 0
            except Exception:  # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1
        cb_ar = capi.ALLREDUCE_FN(_ar)
        cb_ag = capi.ALLGATHER_FN(_ag) if allgather is not None else C.cast(None, capi.ALLGATHER_FN)
        self._keep += [cb_ar, cb_ag]
        check(self.lib.lbfgsb_hip_comm_init_host(self.h, cb_ar, cb_ag, None, rank, nranks))

    # ---- setulb, device-pointer form ----
    @property
    def task_s(self) -> str:
        return task_str(self.task)

    def set_task(self, s: str):
        self.task[:] = pad60(s)

    def wait_stream(self, stream=None):
        """Order the solver's stream after everything queued so far on `stream` (default: torch's
        current stream) without blocking the host (lbfgsb_hip_wait_stream)."""
        if stream is None:
            import torch
            stream = torch.cuda.current_stream().cuda_stream
        check(self.lib.lbfgsb_hip_wait_stream(self.h, C.c_void_p(int(stream))))

    def release_to_stream(self, stream=None):
        """Make `stream` (default: torch's current stream) wait for everything the last call queued on the solver's
        stream -- the trial point of an 'FG...' return included -- without blocking the host
        (lbfgsb_hip_return_event)."""
        if stream is None:
            import torch
            stream = torch.cuda.current_stream().cuda_stream
        check(self.lib.lbfgsb_hip_return_event(self.h, C.c_void_p(int(stream)), 1, None))

    def set_f_device(self, f_dev, stream=None):
        """The objective value as a 0-d / 1-element float64 CUDA tensor produced on `stream` (default: torch's current
        stream): it reaches the host with the next setulb call's own fetch (lbfgsb_hip_f_device) -- no .item()."""
        import torch
        if not (f_dev.is_cuda and f_dev.dtype == torch.float64 and f_dev.numel() == 1):
            raise TypeError("set_f_device needs a one-element float64 CUDA tensor")
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        self._keep_f = f_dev          # (alive until the next call has fetched it)
        check(self.lib.lbfgsb_hip_f_device(self.h, C.c_void_p(f_dev.data_ptr()), C.c_void_p(int(stream)), 1))

    @property
    def stream(self) -> int:
        """the hipStream_t every kernel of this context runs on"""
        return int(self.lib.lbfgsb_hip_get_stream(self.h) or 0)

    # The small caller arrays of a context (task, csave, lsave, isave, dsave, f) live as long as the
    # context and are only ever written in place: their addresses are taken once.  (Per call the wrapper
    # then costs a few microseconds instead of ~20: at n = 1e6 an iteration is 150 us, and the NEW_X
    # return -> re-entry sits on the path during which the device waits for the host.)
    # (The cache is keyed on the array OBJECTS themselves, compared by identity, and keeps them alive: a
    #  replaced array -- sol.isave = saved.copy() in a checkpoint / restore flow -- is always noticed; an id()
    #  alone can be reused by CPython for a new object at another address.)
    _OWN_SPEC = (("f", np.float64, 1), ("task", np.uint8, 60), ("csave", np.uint8, 60), ("lsave", np.int32, 4),
                 ("isave", np.int32, 44), ("dsave", np.float64, 29))

    def _own_ptrs(self):
        arrs = (self.f, self.task, self.csave, self.lsave, self.isave, self.dsave)
        held = getattr(self, "_own_arrs", None)
        if held is None or any(a is not b for a, b in zip(arrs, held)):
            for a, (nm, dt, ln) in zip(arrs, self._OWN_SPEC):
                if not (isinstance(a, np.ndarray) and a.dtype == dt and a.size >= ln and a.flags["C_CONTIGUOUS"]
                        and a.flags["WRITEABLE"]):
                    raise TypeError("DeviceSolver.%s must be a writable contiguous %s array of >= %d elements"
                                    % (nm, np.dtype(dt).name, ln))
            self._own_arrs = arrs
            self._own = tuple(a.ctypes.data for a in arrs)
        return self._own

    @staticmethod
    def _addr(a):
        return a.ctypes.data if isinstance(a, np.ndarray) else a.data_ptr()

    def _before_call(self, first):
        """stream ordering of the caller's tensors (include/lbfgsb_hip.h, "Stream ordering")"""
        if isinstance(first, np.ndarray):
            return
        head = bytes(self.task[:5])
        if head == b"START":
            # the solver runs on its own stream: make sure the caller's tensors are materialised
            import torch
            torch.cuda.synchronize()
        elif head[:2] == b"FG" and not self.same_stream_objective:
            # g (and x) were produced on torch's current stream: the solver's stream must not
            # read them before that work is done
            self.wait_stream()

    def setulb(self, x, l, u, nbd, g, factr: float, pgtol: float, iprint: int = -1) -> str:
        self._before_call(x)
        pf, pt, pc, pl, pi, pd = self._own_ptrs()
        a = self._addr
        check(self.lib.lbfgsb_hip_setulb_dev(self.h, a(x), a(l), a(u), a(nbd), pf, a(g), float(factr),
                                             float(pgtol), pt, int(iprint), pc, pl, pi, pd))
        if self.stream_ordered and bytes(self.task[:2]) == b"FG" and not isinstance(x, np.ndarray):
            self.release_to_stream()   # (the caller's stream may read the trial point from here on)
        return self.task_s

    def setulb_pp(self, xs, l, u, nbd, gs, factr: float, pgtol: float, iprint: int = -1):
        """Ping-pong form (lbfgsb_hip_setulb_dev_pp): xs = (x0, x1), gs = (g0, g1) -- two pairs of
        device buffers, x0 holding the starting point.  -> (task, cur): evaluate f, g at xs[cur] into
        gs[cur] on 'FG...'; xs[cur], gs[cur] are the iterate and its gradient on 'NEW_X' and at the end."""
        self._before_call(xs[0])
        pf, pt, pc, pl, pi, pd = self._own_ptrs()
        a = self._addr
        cur = self._cur
        check(self.lib.lbfgsb_hip_setulb_dev_pp(self.h, a(xs[0]), a(xs[1]), a(l), a(u), a(nbd), pf, a(gs[0]),
                                                a(gs[1]), float(factr), float(pgtol), pt, int(iprint), pc, pl,
                                                pi, pd, self._cur_ref))
        if self.stream_ordered and bytes(self.task[:2]) == b"FG":
            self.release_to_stream()   # (the caller's stream may read the trial point from here on)
        return self.task_s, cur.value

    def minimize(self, x, l, u, nbd, g, fg=None, builtin: int = 0, factr: float = 1e7,
                 pgtol: float = 1e-5, max_iter: int = 0, max_fg: int = 0, iprint: int = -1) -> str:
        """The reference's @todo wrapper (src/lbfgsb.f90:36-37): run the reverse-communication
        loop inside the library.  fg(x_ptr, g_ptr) -> global f evaluates the objective on the
        device (raw device pointers as ints); fg=None uses the built-in objective `builtin`."""
        if not isinstance(x, np.ndarray):
            import torch
            torch.cuda.synchronize()
        cb = C.cast(None, capi.FG_FN)
        failure = []
        if fg is not None:
            def _fg(user, xp, gp):
                # the callback must leave g complete, or ordered before the solver's stream
                # (self.wait_stream() / evaluate on self.stream); a Python exception cannot cross
                # the C frame: it is kept, the evaluation reports NaN -- the line search then
                # fails and the loop ends -- and it is raised again below
                if failure:
                    return float("nan")
                try:
                    val = float(fg(xp, gp))
                    if not self.same_stream_objective:
                        self.wait_stream()
                    return val
                except BaseException as e:   # noqa: BLE001
                    failure.append(e)
                    return float("nan")
            cb = capi.FG_FN(_fg)
            self._keep.append(cb)
        check(self.lib.lbfgsb_hip_minimize(self.h, _p(x), _p(l), _p(u), _p(nbd), _p(g),
                                           float(factr), float(pgtol), int(max_iter), int(max_fg),
                                           int(iprint), cb, None, int(builtin), _p(self.f),
                                           _p(self.task), _p(self.lsave), _p(self.isave),
                                           _p(self.dsave)))
        if failure:
            raise failure[0]
        return self.task_s

    # ---- state exchange / kernels ----
    def export_state(self):
        wa = np.zeros(wa_length(self.n, self.m), self.real)
        iwa = np.zeros(3 * self.n, np.int32)
        check(self.lib.lbfgsb_hip_export_state(self.h, _p(wa), _p(iwa)))
        return wa, iwa

    def import_state(self, wa, iwa, isave):
        wa = np.ascontiguousarray(wa, self.real)
        iwa = np.ascontiguousarray(iwa, np.int32)
        isave = np.ascontiguousarray(isave, np.int32)
        check(self.lib.lbfgsb_hip_import_state(self.h, _p(wa), _p(iwa), _p(isave)))

    def projgr(self, x, l, u, nbd, g) -> float:
        out = np.zeros(1)
        check(self.lib.lbfgsb_hip_projgr(self.h, _p(x), _p(l), _p(u), _p(nbd), _p(g), _p(out)))
        return float(out[0])

    # ---- routine doors (include/lbfgsb_hip.h "Routine doors"): one routine of the reference each, on the
    #      state of the context (import_state / export_state); vectors are device tensors ----
    def r_vec_sub(self, a, b, out):
        """out = a - b (level-1 door; src/lbfgsb.f90:720-722, :812-816)"""
        check(self.lib.lbfgsb_hip_vec_sub(self.h, _p(a), _p(b), _p(out)))

    def r_vec_scale(self, alpha, v):
        """v = alpha v (dscal, :822)"""
        check(self.lib.lbfgsb_hip_vec_scale(self.h, float(alpha), _p(v)))

    def r_dot(self, a, b):
        """a'b in fp64, summed over the context's ranks (ddot, :816 / :2196 / :2244 / :2335)"""
        out = np.zeros(1, np.float64)
        check(self.lib.lbfgsb_hip_dot(self.h, _p(a), _p(b), _p(out)))
        return float(out[0])

    def r_active(self, x, l, u, nbd):
        """active :965 -- x projected in place; -> (prjctd, cnstnd, boxed)"""
        out = np.zeros(3, np.int32)
        check(self.lib.lbfgsb_hip_active(self.h, _p(x), _p(l), _p(u), _p(nbd), _p(out)))
        return bool(out[0]), bool(out[1]), bool(out[2])

    def r_errclb(self, l, u, nbd, factr, task=b"START"):
        """errclb :1601 -- -> (task string, info, k)"""
        t = np.frombuffer(task.ljust(60), dtype="S1").copy()
        info, k = np.zeros(1, np.int32), np.zeros(1, np.int64)
        check(self.lib.lbfgsb_hip_errclb(self.h, _p(l), _p(u), _p(nbd), float(factr), _p(t), _p(info), _p(k)))
        return t.tobytes().decode().rstrip(), int(info[0]), int(k[0])

    def r_cauchy(self, x, l, u, nbd, g, theta, col, head, sbgnrm, xcp_out=None):
        """cauchy :1157 -- -> (nseg, info); xcp in the state's z (and in xcp_out)"""
        nseg, info = np.zeros(1, np.int32), np.zeros(1, np.int32)
        check(self.lib.lbfgsb_hip_cauchy(self.h, _p(x), _p(l), _p(u), _p(nbd), _p(g), float(theta), int(col),
                                         int(head), float(sbgnrm), _p(xcp_out) if xcp_out is not None else None,
                                         _p(nseg), _p(info)))
        return int(nseg[0]), int(info[0])

    def r_freev(self, iter_, cnstnd, updatd):
        """freev :1980 -- -> (nfree, nenter, ileave, wrk)"""
        o = np.zeros(3, np.int64)
        wrk = np.zeros(1, np.int32)
        check(self.lib.lbfgsb_hip_freev(self.h, int(iter_), int(bool(cnstnd)), int(bool(updatd)), _p(o[0:1]),
                                        _p(o[1:2]), _p(o[2:3]), _p(wrk)))
        return int(o[0]), int(o[1]), int(o[2]), bool(wrk[0])

    def r_formk(self, col, head, theta):
        """formk :1681 (inner products from scratch) -- -> info; WN1, WN in the state"""
        info = np.zeros(1, np.int32)
        check(self.lib.lbfgsb_hip_formk(self.h, int(col), int(head), float(theta), _p(info)))
        return int(info[0])

    def r_cmprlb(self, x, g, theta, col, head, cnstnd, r_out):
        """cmprlb :1548 -- r scattered to its rows in r_out -> info"""
        info = np.zeros(1, np.int32)
        check(self.lib.lbfgsb_hip_cmprlb(self.h, _p(x), _p(g), float(theta), int(col), int(head),
                                         int(bool(cnstnd)), _p(r_out), _p(info)))
        return int(info[0])

    def r_subsm(self, x, l, u, nbd, g, r_in, theta, col, head, xhat_out=None):
        """subsm :2676 -- -> (iword, info); the subspace minimiser in the state's z (and xhat_out)"""
        iword, info = np.zeros(1, np.int32), np.zeros(1, np.int32)
        check(self.lib.lbfgsb_hip_subsm(self.h, _p(x), _p(l), _p(u), _p(nbd), _p(g), _p(r_in), float(theta),
                                        int(col), int(head), _p(xhat_out) if xhat_out is not None else None,
                                        _p(iword), _p(info)))
        return int(iword[0]), int(info[0])

    def r_lnsrlb(self, x, l, u, nbd, g, f, sc, ic, task, csave, isave2, dsave13):
        """lnsrlb :2174 -- sc (8 doubles), ic (7 int32), task / csave (S1[60]), isave2, dsave13 in / out"""
        check(self.lib.lbfgsb_hip_lnsrlb(self.h, _p(x), _p(l), _p(u), _p(nbd), _p(g), float(f), _p(sc), _p(ic),
                                         _p(task), _p(csave), _p(isave2), _p(dsave13)))

    def r_matupd(self, g, stp, dr, dtd, ip):
        """mainlb :812-824 + matupd :2291 -- ip = (iupdat, col, head, itail) int32 in / out -> theta"""
        th = np.zeros(1)
        check(self.lib.lbfgsb_hip_matupd(self.h, _p(g), float(stp), float(dr), float(dtd), _p(ip), _p(th)))
        return float(th[0])

    def set_w(self, ws: np.ndarray, wy: np.ndarray):
        """ws, wy: host arrays of shape (m, n) C-order == Fortran (n, m) column-major."""
        ws = np.ascontiguousarray(ws, self.real)
        wy = np.ascontiguousarray(wy, self.real)
        assert ws.shape == (self.m, self.n) and wy.shape == (self.m, self.n)
        check(self.lib.lbfgsb_hip_set_w(self.h, _p(ws), _p(wy)))

    def set_iwhere(self, iwhere: np.ndarray):
        iw = np.ascontiguousarray(iwhere, np.int32)
        assert iw.shape == (self.n,)
        check(self.lib.lbfgsb_hip_set_iwhere(self.h, _p(iw)))

    def formk_gram(self, col: int, head: int = 1) -> np.ndarray:
        """formk's inner products from scratch (layout: include/lbfgsb_hip.h)"""
        out = np.zeros(2 * col * col + col)
        check(self.lib.lbfgsb_hip_formk_gram(self.h, col, head, _p(out)))
        return out

    def wtv(self, v, col: int, head: int = 1) -> np.ndarray:
        out = np.zeros(2 * col)
        check(self.lib.lbfgsb_hip_wtv(self.h, _p(v), col, head, _p(out)))
        return out

    def wtv_launch(self, v, col: int, head: int = 1):
        check(self.lib.lbfgsb_hip_wtv_launch_only(self.h, _p(v), col, head))

    def wtv_time(self, v, col: int, head: int = 1, reps: int = 20) -> float:
        """average milliseconds of one W'v kernel launch (hipEvents on the solver's stream)"""
        out = np.zeros(1)
        check(self.lib.lbfgsb_hip_wtv_time(self.h, _p(v), col, head, reps, _p(out)))
        return float(out[0])

    def kernel_time(self, which: int, x, g, col: int, head: int = 1, reps: int = 20) -> float:
        """average ms per launch of an in-iteration kernel: 0 = cmprlb_wtv, 1 = formk gram"""
        out = np.zeros(1)
        check(self.lib.lbfgsb_hip_kernel_time(self.h, which, _p(x), _p(g), col, head, reps, _p(out)))
        return float(out[0])

    def sync(self):
        check(self.lib.lbfgsb_hip_sync(self.h))

    def objective(self, kind: int, x, g, deferred: bool = False):
        """Built-in objective on the solver's stream.  deferred=True leaves f on the device: the
        next setulb() call (the FG re-entry) fetches it with its own sums and stores it in
        self.f -- one host sync less per evaluation; returns None then."""
        if deferred:
            check(self.lib.lbfgsb_hip_objective(self.h, kind, self._addr(x), self._addr(g), None))
            return None
        out = np.zeros(1)
        check(self.lib.lbfgsb_hip_objective(self.h, kind, _p(x), _p(g), _p(out)))
        return float(out[0])

    def pass_clock(self, enable: int):
        """In-run hipEvent clocks of the three passes over W (see lbfgsb_hip_pass_clock):
        enable 1 = reset and start, 0 = stop, -1 = read.  -> {name: (ms_total, launches)}"""
        ms = (C.c_double * 3)()
        cnt = (C.c_int64 * 3)()
        check(self.lib.lbfgsb_hip_pass_clock(self.h, int(enable), ms, cnt))
        names = ("cmprlb_wtv", "update_scan", "subsm_update")
        return {nm: (ms[k], cnt[k]) for k, nm in enumerate(names)}

    def path_counts(self):
        """(subspace steps by the two-pass closed form, by the three-pass route, Cauchy walks
        served by the breakpoints the update pass handed over)"""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        check(self.lib.lbfgsb_hip_path_counts(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return int(a.value), int(b.value), int(c.value)

    def uniform_bounds(self) -> int:
        """bit 0 / 1 / 2: l / u / nbd hold one value each and are read as constants by the passes over W"""
        c = C.c_int32()
        check(self.lib.lbfgsb_hip_uniform_bounds(self.h, C.byref(c)))
        return int(c.value)

    def comm_info(self):
        """(nranks, rank, kind) of the context's communicator; kind 0 none, 1 RCCL (the communicator's own
        ncclCommCount / ncclCommUserRank), 2 host callbacks"""
        a, b, k = C.c_int32(), C.c_int32(), C.c_int32()
        check(self.lib.lbfgsb_hip_comm_info(self.h, C.byref(a), C.byref(b), C.byref(k)))
        return int(a.value), int(b.value), int(k.value)

    def compact_stats(self):
        """option compact_w: (packs, unpacks, packed now, eligible) -- lbfgsb_hip_compact_stats"""
        a, b = C.c_int64(0), C.c_int64(0)
        c, d = C.c_int32(0), C.c_int32(0)
        check(self.lib.lbfgsb_hip_compact_stats(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return a.value, b.value, c.value, d.value

    def collective_time(self, reps: int = 1000):
        """(median_us, min_us) of one host sync of the iteration by itself (lbfgsb_hip_collective_time); a
        collective: every rank calls it"""
        a, b = C.c_double(0.0), C.c_double(0.0)
        check(self.lib.lbfgsb_hip_collective_time(self.h, int(reps), C.byref(a), C.byref(b)))
        return a.value, b.value

    def host_gap(self):
        """(seconds, stretches) the device waited for the host's 2m x 2m algebra between the two passes"""
        a, b = C.c_double(), C.c_int64()
        check(self.lib.lbfgsb_hip_host_gap(self.h, C.byref(a), C.byref(b)))
        return float(a.value), int(b.value)

    def host_segments(self):
        """accumulated host seconds of the stretch between the two passes, by milestone (lbfgsb_hip_host_segments)"""
        a = (C.c_double * 5)()
        check(self.lib.lbfgsb_hip_host_segments(self.h, a))
        return [float(v) for v in a]

    def defer_stats(self):
        """(line-search set-ups whose sums were deferred, of those: requests that had to be re-issued)"""
        a, b = C.c_int64(), C.c_int64()
        check(self.lib.lbfgsb_hip_defer_stats(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def tie_splits(self) -> int:
        """setulb calls so far whose Cauchy walk ended inside a group of equal breakpoints"""
        c = C.c_int64()
        check(self.lib.lbfgsb_hip_tie_splits(self.h, C.byref(c)))
        return int(c.value)

    def stats(self):
        a, b, c, w = C.c_int64(), C.c_int64(), C.c_int64(), C.c_double()
        check(self.lib.lbfgsb_hip_stats(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(w)))
        nc, nb = C.c_int64(), C.c_int64()
        check(self.lib.lbfgsb_hip_comm_stats(self.h, C.byref(nc), C.byref(nb)))
        fs = C.c_int64()
        check(self.lib.lbfgsb_hip_freev_skipped(self.h, C.byref(fs)))
        sr = C.c_int64()
        check(self.lib.lbfgsb_hip_skip_stats(self.h, C.byref(sr)))
        rf = C.c_int64()
        check(self.lib.lbfgsb_hip_refresh_count(self.h, C.byref(rf)))
        return dict(launches=a.value, syncs=b.value, cauchy_fullsorts=c.value, wait_seconds=w.value,
                    collectives=nc.value, collective_bytes=nb.value, freev_skipped=fs.value,
                    skip_scans_reused=sr.value, refreshes=rf.value)

"""CPU tests of the PRODUCT's host logic (no GPU): the 2m x 2m algebra that stays on the host
(lbfgsb_amd/csrc/host_dense.hpp: dpofa, dtrsl, bmv, formt, dcsrch, hpsolb -- reference
src/lbfgsb_linpack_module.f90:30,87 and src/lbfgsb.f90:1057,1926,2942,2079) against the oracle,
bit for bit, and the Fortran edit-descriptor emulation of report.hpp."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def libs(oracle_built):
    out = os.path.join(HERE, "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libhost_shim.so")
    # LBFGSB_TEST_SANITIZE=1: the same tests with UBSan on the host algebra (CPU build only; GPU
    # sanitizers are not available on the pool)
    san = (["-fsanitize=undefined", "-fno-sanitize-recover=undefined", "-g"]
           if os.environ.get("LBFGSB_TEST_SANITIZE") == "1" else [])
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared"] + san +
                          [os.path.join(HERE, "host_shim.cpp"), "-o", so])
    hd = C.CDLL(so)
    orc = oracle_built.Engine("oracle").lib
    return hd, orc


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def spd(rng, n, lda):
    a = rng.standard_normal((n, n))
    a = a @ a.T + n * np.eye(n)
    out = np.zeros((lda, lda))
    out[:n, :n] = a
    return np.asfortranarray(out)


def test_dpofa_dtrsl_bit_identical(libs):
    hd, orc = libs
    rng = np.random.default_rng(1)
    for n, lda in ((1, 3), (4, 4), (7, 10), (20, 40)):
        a = spd(rng, n, lda)
        a1, a2 = a.copy(order="F"), a.copy(order="F")
        info = C.c_int(0)
        r1 = hd.hd_dpofa(_p(a1), lda, n)
        orc.lbo_dpofa(_p(a2), lda, n, C.byref(info))
        assert r1 == info.value == 0
        assert a1.tobytes() == a2.tobytes()
        for job in (0, 1, 10, 11):
            t = np.asfortranarray(np.triu(a1) if job % 10 else np.tril(a1.T))
            b = rng.standard_normal(n)
            b1, b2 = b.copy(), b.copy()
            r1 = hd.hd_dtrsl(_p(t), lda, n, _p(b1), job)
            orc.lbo_dtrsl(_p(t), lda, n, _p(b2), job, C.byref(info))
            assert r1 == info.value == 0 and b1.tobytes() == b2.tobytes()
    # failure codes: not positive definite -> order of the failing minor; zero diagonal -> its index
    a = np.asfortranarray(np.diag([1.0, -1.0, 1.0]))
    info = C.c_int(0)
    orc.lbo_dpofa(_p(a.copy(order="F")), 3, 3, C.byref(info))
    assert hd.hd_dpofa(_p(a.copy(order="F")), 3, 3) == info.value == 2
    t = np.asfortranarray(np.diag([1.0, 0.0, 1.0]))
    b = np.ones(3)
    orc.lbo_dtrsl(_p(t), 3, 3, _p(b.copy()), 11, C.byref(info))
    assert hd.hd_dtrsl(_p(t), 3, 3, _p(b.copy()), 11) == info.value == 2


def test_formt_bmv_bit_identical(libs):
    hd, orc = libs
    rng = np.random.default_rng(2)
    for m, col in ((5, 1), (5, 5), (10, 7), (20, 20)):
        s = rng.standard_normal((50, m))
        y = s * (1.0 + rng.random((50, m))) + 0.1 * rng.standard_normal((50, m))
        sy = np.asfortranarray(s.T @ y)
        ss = np.asfortranarray(s.T @ s)
        theta = 1.7
        wt1, wt2 = np.zeros((m, m), order="F"), np.zeros((m, m), order="F")
        info = C.c_int(0)
        r1 = hd.hd_formt(m, _p(wt1), _p(sy), _p(ss), col, C.c_double(theta))
        orc.lbo_formt(m, _p(wt2), _p(sy), _p(ss), col, C.c_double(theta), C.byref(info))
        assert r1 == info.value
        assert wt1.tobytes() == wt2.tobytes()
        if r1 != 0:
            continue
        v = rng.standard_normal(2 * col)
        p1, p2 = np.zeros(2 * col), np.zeros(2 * col)
        r1 = hd.hd_bmv(m, _p(sy), _p(wt1), col, _p(v), _p(p1))
        orc.lbo_bmv(m, _p(sy), _p(wt2), col, _p(v), _p(p2), C.byref(info))
        assert r1 == info.value == 0 and p1.tobytes() == p2.tobytes()


def test_dcsrch_bit_identical_sequences(libs):
    """Drive both line searches on the same 1-D functions until they terminate."""
    hd, orc = libs
    funcs = [
        (lambda s: (s - 1.3) ** 2 - 1.69, lambda s: 2 * (s - 1.3)),
        (lambda s: -s / (s * s + 2.0), lambda s: (s * s - 2.0) / (s * s + 2.0) ** 2),
        (lambda s: np.exp(-s) + 0.05 * s * s - 1.0, lambda s: -np.exp(-s) + 0.1 * s),
    ]
    for phi, dphi in funcs:
        for stp0, stpmax in ((1.0, 1e10), (0.01, 5.0), (4.0, 4.0)):
            st = []
            for which in (0, 1):
                task = np.frombuffer(b"START".ljust(60), dtype=np.uint8).copy()
                isave = np.zeros(2, np.int32)
                dsave = np.zeros(13)
                stp = C.c_double(stp0)
                f, g = phi(0.0), dphi(0.0)
                trace = []
                for _ in range(60):
                    if which == 0:
                        hd.hd_dcsrch(C.c_double(f), C.c_double(g), C.byref(stp), C.c_double(1e-3),
                                     C.c_double(0.9), C.c_double(0.1), C.c_double(0.0),
                                     C.c_double(stpmax), _p(task), _p(isave), _p(dsave))
                    else:
                        ff, gg = C.c_double(f), C.c_double(g)
                        orc.lbo_dcsrch(C.byref(ff), C.byref(gg), C.byref(stp), C.c_double(1e-3),
                                       C.c_double(0.9), C.c_double(0.1), C.c_double(0.0),
                                       C.c_double(stpmax), _p(task), _p(isave), _p(dsave))
                    t = bytes(task.tobytes()).rstrip()
                    trace.append((t, stp.value, isave.tobytes(), dsave.tobytes()))
                    if not t.startswith(b"FG"):
                        break
                    f, g = phi(stp.value), dphi(stp.value)
                st.append(trace)
            assert st[0] == st[1]
            assert st[0][-1][0].startswith((b"CONV", b"WARN"))


def test_hpsolb_bit_identical_pop_order(libs):
    """hpsolb (src/lbfgsb.f90:2079-2157) as the product replays it for walks that end inside a group
    of equal breakpoints: heap build + pops over keys with MANY ties must leave t and iorder exactly
    as the oracle's restatement does after every pop -- with 32-bit and with 64-bit row numbers."""
    hd, orc = libs
    rng = np.random.default_rng(7)
    for n in (1, 2, 3, 17, 64, 257, 1000):
        vals = rng.integers(0, max(2, n // 6), n).astype(np.float64) * 0.125   # groups of equal keys
        for width in (32, 64):
            t1, t2 = vals.copy(), vals.copy()
            io_o = np.arange(1, n + 1, dtype=np.int32)
            io_p = (np.arange(1, n + 1, dtype=np.uint32) if width == 32
                    else np.arange(1, n + 1, dtype=np.int64) + (1 << 33))
            fn = hd.hd_hpsolb32 if width == 32 else hd.hd_hpsolb64
            for k, nleft in enumerate(range(n, 0, -1)):
                orc.lbo_hpsolb(nleft, _p(t1), _p(io_o), 0 if k == 0 else 1)
                fn(C.c_int64(nleft), _p(t2), _p(io_p), 0 if k == 0 else 1)
                assert t1.tobytes() == t2.tobytes()
                got = io_p.astype(np.int64) - (0 if width == 32 else (1 << 33))
                assert np.array_equal(got, io_o.astype(np.int64)), (n, width, k)
            assert np.all(np.diff(t1[::-1]) >= 0)      # popped in ascending order of t


def test_fortran_edit_descriptors(libs):
    """1p,dW.D / 1p,eW.D as the reference's transcripts show them (test/OUTPUTS/output_90_1)."""
    hd, _ = libs

    def fmt(v, w, d, letter):
        buf = C.create_string_buffer(64)
        hd.hd_fmt(C.c_double(v), w, d, ord(letter), buf, 64)
        return buf.value.decode()
    assert fmt(3460.0, 12, 5, "D") == " 3.46000D+03"
    assert fmt(103.0, 12, 5, "D") == " 1.03000D+02"
    assert fmt(2.220446049250313e-16, 10, 3, "D") == " 2.220D-16"
    assert fmt(1.083490083e-09, 10, 3, "D") == " 1.083D-09"
    assert fmt(0.012, 7, 1, "D") == "1.2D-02"
    assert fmt(1.0e-3, 10, 3, "E") == " 1.000E-03"
    assert fmt(-5.5, 12, 5, "D") == "-5.50000D+00"

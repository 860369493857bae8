"""ctypes wrapper of oracle/ldlt_oracle.c (see its header for what it restates and how it is pinned)
plus readers for the boundary recordings made from the reference (oracle/ref_driver.cpp --record).
TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "lib", "libkkt_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.kkt_oracle_factor_solve.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double,
                                                 C.c_double, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                                 C.POINTER(C.c_int)]
    return _LIB


def factor_solve(n, row, col, val, rhs=None, base=1, u=1e-8, small=1e-20):
    """returns (x, num_neg, num_zero, num_two); rhs may be (n,) or (nrhs, n)."""
    row = np.ascontiguousarray(row, dtype=np.int32); col = np.ascontiguousarray(col, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float64)
    if rhs is None:
        x = np.zeros((0, n)); nrhs = 0
    else:
        x = np.array(rhs, dtype=np.float64, copy=True).reshape(-1, n); nrhs = x.shape[0]
    neg, zero, two = C.c_int(0), C.c_int(0), C.c_int(0)
    _lib().kkt_oracle_factor_solve(int(n), int(val.shape[0]), row.ctypes.data, col.ctypes.data, val.ctypes.data, int(base),
                                   float(u), float(small), nrhs, x.ctypes.data if nrhs else None, C.byref(neg), C.byref(zero), C.byref(two))
    out = x[0] if (rhs is not None and np.ndim(rhs) == 1) else x
    return out, neg.value, zero.value, two.value


def read_kktrec(path):
    """parse a boundary recording (format written by RecordingSolverInterface in oracle/ref_driver.cpp).
    Returns dict(dim, nnz, fmt, ia, ja, calls=[dict(new_matrix, check, required_neg, status, neg, a, rhs, sol)]);
    `a` of a call with new_matrix == 0 is the previous call's array."""
    buf = open(path, "rb").read()
    assert buf[:8] == b"KKTREC1\n"
    off = 8
    out = dict(calls=[])
    a_prev = None
    while off < len(buf):
        hdr = np.frombuffer(buf, dtype=np.int32, count=8, offset=off); off += 32
        if hdr[0] == 0:
            dim, nnz, fmt, nia = int(hdr[1]), int(hdr[2]), int(hdr[3]), int(hdr[4])
            ia = np.frombuffer(buf, dtype=np.int32, count=nia, offset=off).copy(); off += 4 * nia
            ja = np.frombuffer(buf, dtype=np.int32, count=nnz, offset=off).copy(); off += 4 * nnz
            out.update(dim=dim, nnz=nnz, fmt=fmt, ia=ia, ja=ja)
        else:
            dim, nnz, nrhs, newm, check, req, status = (int(v) for v in hdr[1:8])
            neg = int(np.frombuffer(buf, dtype=np.int32, count=1, offset=off)[0]); off += 4
            if newm:
                a_prev = np.frombuffer(buf, dtype=np.float64, count=nnz, offset=off).copy(); off += 8 * nnz
            rhs = np.frombuffer(buf, dtype=np.float64, count=dim * nrhs, offset=off).copy(); off += 8 * dim * nrhs
            sol = np.frombuffer(buf, dtype=np.float64, count=dim * nrhs, offset=off).copy(); off += 8 * dim * nrhs
            out["calls"].append(dict(new_matrix=newm, check=check, required_neg=req, status=status, neg=neg, a=a_prev,
                                     rhs=rhs.reshape(nrhs, dim), sol=sol.reshape(nrhs, dim)))
    return out


def rec_triplets(rec):
    """(row, col) 1-based triplets of a recording, whatever format the recorded backend asked for
    (EMatrixFormat, reference IpSparseSymLinearSolverInterface.hpp:102-114: 0 triplet, 1/2 CSR upper 0/1-offset)."""
    fmt = rec["fmt"]
    if fmt == 0:
        return rec["ia"], rec["ja"]
    base = 0 if fmt in (1, 3) else 1
    ia = rec["ia"] - base
    rows = np.repeat(np.arange(rec["dim"], dtype=np.int32), np.diff(ia)) + 1
    return rows, rec["ja"] - base + 1

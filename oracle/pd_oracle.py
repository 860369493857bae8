"""CPU restatement of the reference's 8-block primal-dual algebra -- TEST INFRASTRUCTURE ONLY: imported by tests/,
never by the product.  Plain numpy, dense, for small systems.

  * read_pdrec(): the recordings `oracle/ref_driver --record-pd` makes at the PDSystemSolver boundary of the UNMODIFIED reference
    (what PDFullSpaceSolver::Solve was given and what it returned); fixtures in tests/golden/*.pdrec.
  * k8_dense(): the matrix of IpPDSystemSolver.hpp:24-49 with the perturbations of IpPDFullSpaceSolver.cpp:666-793 (ComputeResiduals).
  * solve_once(): SolveOnce restated (IpPDFullSpaceSolver.cpp:377-664): bound rows eliminated into the right-hand side (:418-424,
    ExpansionMatrix::AddMSinvZ), the 4-block augmented system W + Sigma_x + delta_x I ... solved densely, bound blocks recovered
    (:653-659, ExpansionMatrix::SinvBlrmZMTdBr).
  * residual(): ComputeResiduals restated block by block (:693-760) and the ratio of ComputeResidualRatio (:795-820).

Pinned (tests/test_pd_oracle.py, CPU): on every recorded call the restated solve_once() agrees with a dense solve of k8_dense(), and the
vector the reference returned (after its iterative refinement) satisfies k8_dense() . res = alpha . rhs to the reference's own
residual_ratio_max."""
from __future__ import annotations

import struct

import numpy as np


def read_pdrec(path):
    raw = open(path, "rb").read()
    assert raw[:7] == b"PDREC1\n", "not a PDREC1 file"
    off = 7
    recs = []

    def take(fmt, count):
        nonlocal off
        dt = np.dtype(fmt)
        a = np.frombuffer(raw, dtype=dt, count=count, offset=off).copy()
        off += count * dt.itemsize
        return a

    while off < len(raw):
        h = take("<i4", 16)
        nx, ns, nc, nd, nxl, nxu, nsl, nsu, nW, nJc, nJd, inexact, improve, ok = (int(v) for v in h[:14])
        sc = take("<f8", 6)
        r = dict(nx=nx, ns=ns, nc=nc, nd=nd, allow_inexact=bool(inexact), improve_solution=bool(improve), ok=bool(ok),
                 alpha=float(sc[0]), beta=float(sc[1]), deltas=sc[2:6].copy())
        r["ixl"], r["ixu"], r["isl"], r["isu"] = take("<i4", nxl), take("<i4", nxu), take("<i4", nsl), take("<i4", nsu)
        for name, n in (("W", nW), ("Jc", nJc), ("Jd", nJd)):
            r[name] = (take("<i4", n) - 1, take("<i4", n) - 1, take("<f8", n))          # 0-based (row, col, value)
        for name, n in (("zl", nxl), ("zu", nxu), ("vl", nsl), ("vu", nsu), ("sxl", nxl), ("sxu", nxu), ("ssl", nsl), ("ssu", nsu),
                        ("sigma_x", nx), ("sigma_s", ns)):
            r[name] = take("<f8", n)
        n8 = nx + ns + nc + nd + nxl + nxu + nsl + nsu
        r["rhs"], r["res_in"], r["res_out"] = take("<f8", n8), take("<f8", n8), take("<f8", n8)
        recs.append(r)
    return recs


def offsets(r):
    return np.cumsum([0, r["nx"], r["ns"], r["nc"], r["nd"], len(r["ixl"]), len(r["ixu"]), len(r["isl"]), len(r["isu"])])


def split(r, v):
    o = offsets(r)
    return [v[o[i]:o[i + 1]] for i in range(8)]


def _dense(trip, m, n, sym=False):
    M = np.zeros((m, n))
    rr, cc, vv = trip
    np.add.at(M, (rr, cc), vv)
    if sym:                                     # SymTMatrix: one triangle stored, off-diagonal entries count for both
        offd = rr != cc
        np.add.at(M, (cc[offd], rr[offd]), vv[offd])
    return M


def matrices(r):
    return _dense(r["W"], r["nx"], r["nx"], sym=True), _dense(r["Jc"], r["nc"], r["nx"]), _dense(r["Jd"], r["nd"], r["nx"])


def k8_dense(r, deltas=None):
    """IpPDSystemSolver.hpp:24-49 with delta_x, delta_s on the (x,x), (s,s) blocks and -delta_c, -delta_d on (c,c), (d,d)."""
    dx, ds, dc, dd = r["deltas"] if deltas is None else deltas
    nx, ns, nc, nd = r["nx"], r["ns"], r["nc"], r["nd"]
    W, Jc, Jd = matrices(r)
    o = offsets(r)
    X, S, C, D, ZL, ZU, VL, VU = [slice(o[i], o[i + 1]) for i in range(8)]
    E = lambda n, idx: np.eye(n)[:, idx]
    Pxl, Pxu, Pdl, Pdu = E(nx, r["ixl"]), E(nx, r["ixu"]), E(ns, r["isl"]), E(ns, r["isu"])
    K = np.zeros((o[-1], o[-1]))
    K[X, X] = W + dx * np.eye(nx); K[X, C] = Jc.T; K[X, D] = Jd.T; K[X, ZL] = -Pxl; K[X, ZU] = Pxu
    K[S, S] = ds * np.eye(ns); K[S, D] = -np.eye(ns); K[S, VL] = -Pdl; K[S, VU] = Pdu
    K[C, X] = Jc; K[C, C] = -dc * np.eye(nc)
    K[D, X] = Jd; K[D, S] = -np.eye(nd); K[D, D] = -dd * np.eye(nd)
    K[ZL, X] = np.diag(r["zl"]) @ Pxl.T; K[ZL, ZL] = np.diag(r["sxl"])
    K[ZU, X] = -np.diag(r["zu"]) @ Pxu.T; K[ZU, ZU] = np.diag(r["sxu"])
    K[VL, S] = np.diag(r["vl"]) @ Pdl.T; K[VL, VL] = np.diag(r["ssl"])
    K[VU, S] = -np.diag(r["vu"]) @ Pdu.T; K[VU, VU] = np.diag(r["ssu"])
    return K


def solve_once(r, rhs, deltas=None):
    """SolveOnce restated: reduce, 4-block solve, expand.  Returns the 8-block solution."""
    dx, ds, dc, dd = r["deltas"] if deltas is None else deltas
    nx, ns, nc, nd = r["nx"], r["ns"], r["nc"], r["nd"]
    W, Jc, Jd = matrices(r)
    bx, bs, bc, bd, bzl, bzu, bvl, bvu = split(r, rhs)
    ax = bx.copy(); np.add.at(ax, r["ixl"], bzl / r["sxl"]); np.subtract.at(ax, r["ixu"], bzu / r["sxu"])        # :418-420
    as_ = bs.copy(); np.add.at(as_, r["isl"], bvl / r["ssl"]); np.subtract.at(as_, r["isu"], bvu / r["ssu"])      # :422-424
    n4 = nx + ns + nc + nd
    A = np.zeros((n4, n4))
    X, S, C, D = slice(0, nx), slice(nx, nx + ns), slice(nx + ns, nx + ns + nc), slice(nx + ns + nc, n4)
    A[X, X] = W + np.diag(r["sigma_x"]) + dx * np.eye(nx); A[S, S] = np.diag(r["sigma_s"]) + ds * np.eye(ns)
    A[C, X] = Jc; A[X, C] = Jc.T; A[D, X] = Jd; A[X, D] = Jd.T; A[D, S] = -np.eye(nd); A[S, D] = -np.eye(ns)
    A[C, C] = -dc * np.eye(nc); A[D, D] = -dd * np.eye(nd)
    sol4 = np.linalg.solve(A, np.concatenate([ax, as_, bc, bd]))
    sx, ss = sol4[X], sol4[S]
    zl = (bzl - r["zl"] * sx[r["ixl"]]) / r["sxl"]; zu = (bzu + r["zu"] * sx[r["ixu"]]) / r["sxu"]                  # :653-656
    vl = (bvl - r["vl"] * ss[r["isl"]]) / r["ssl"]; vu = (bvu + r["vu"] * ss[r["isu"]]) / r["ssu"]
    return np.concatenate([sol4, zl, zu, vl, vu])


def residual(r, rhs, res, deltas=None):
    """ComputeResiduals restated block by block; returns (resid, ratio of ComputeResidualRatio)."""
    dx, ds, dc, dd = r["deltas"] if deltas is None else deltas
    W, Jc, Jd = matrices(r)
    bx, bs, bc, bd, bzl, bzu, bvl, bvu = split(r, rhs)
    x, s, yc, yd, zl, zu, vl, vu = split(r, res)
    rx = W @ x + Jc.T @ yc + Jd.T @ yd; np.subtract.at(rx, r["ixl"], zl); np.add.at(rx, r["ixu"], zu); rx = rx + dx * x - bx
    rs = np.zeros(r["ns"]); np.add.at(rs, r["isu"], vu); np.subtract.at(rs, r["isl"], vl); rs = rs - yd - bs + ds * s
    rc = Jc @ x - dc * yc - bc
    rd = Jd @ x - s - bd - dd * yd
    rzl = zl * r["sxl"] + r["zl"] * x[r["ixl"]] - bzl; rzu = zu * r["sxu"] - r["zu"] * x[r["ixu"]] - bzu
    rvl = vl * r["ssl"] + r["vl"] * s[r["isl"]] - bvl; rvu = vu * r["ssu"] - r["vu"] * s[r["isu"]] - bvu
    resid = np.concatenate([rx, rs, rc, rd, rzl, rzu, rvl, rvu])
    nr, ns_, nd_ = np.abs(rhs).max(initial=0.0), np.abs(res).max(initial=0.0), np.abs(resid).max(initial=0.0)
    ratio = nd_ if nr + ns_ == 0.0 else nd_ / (min(ns_, 1e6 * nr) + nr)
    return resid, ratio

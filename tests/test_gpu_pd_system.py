"""-m gpu: the 8-block primal-dual kernels of the C ABI (mi355x_kkt_pd_*) against a dense numpy statement of
the reference's formulas: SolveOnce's reduction / expansion (IpPDFullSpaceSolver.cpp:418-424, :653-659), ComputeResiduals
(:666-793) and the norms of ComputeResidualRatio (:795-820).  The matrices reach the device through the value assembly
(segments W | D_x | D_s | J_c | D_c | J_d | -I | D_d, the layout of IpStdAugSystemSolver.cpp:263-298)."""
import numpy as np
import pytest

import ipopt_amd

pytestmark = pytest.mark.gpu


def build(seed, nx=60, nc=14, nd=9, dens=0.12):
    rng = np.random.default_rng(seed)
    ns = nd
    # W: symmetric, lower triplets incl. a few duplicates; J_c, J_d dense-ish random patterns
    Wm = np.zeros((nx, nx)); wr, wc, wv = [], [], []
    for _ in range(int(dens * nx * nx / 2) + nx):
        i, j = sorted(rng.integers(0, nx, 2))[::-1]
        v = rng.standard_normal(); wr.append(i); wc.append(j); wv.append(v)
        Wm[i, j] += v
        if i != j:
            Wm[j, i] += v
    def rect(m):
        M = np.zeros((m, nx)); r, c, v = [], [], []
        for i in range(m):
            for j in rng.choice(nx, size=max(2, int(dens * nx)), replace=False):
                x = rng.standard_normal(); r.append(i); c.append(j); v.append(x); M[i, j] += x
        return M, np.array(r), np.array(c), np.array(v)
    Jc, jcr, jcc, jcv = rect(nc)
    Jd, jdr, jdc, jdv = rect(nd)
    pick = lambda n, k: np.sort(rng.choice(n, size=k, replace=False)).astype(np.int32)
    ixl, ixu, isl, isu = pick(nx, nx // 2), pick(nx, nx // 3), pick(ns, ns // 2 + 1), pick(ns, ns // 3 + 1)
    pos = lambda k: rng.uniform(0.1, 2.0, k)
    data = dict(zl=pos(len(ixl)), zu=pos(len(ixu)), vl=pos(len(isl)), vu=pos(len(isu)), sxl=pos(len(ixl)), sxu=pos(len(ixu)), ssl=pos(len(isl)), ssu=pos(len(isu)))
    return dict(nx=nx, ns=ns, nc=nc, nd=nd, W=Wm, wt=(np.array(wr), np.array(wc), np.array(wv)), Jc=Jc, jct=(jcr, jcc, jcv), Jd=Jd, jdt=(jdr, jdc, jdv),
                ixl=ixl, ixu=ixu, isl=isl, isu=isu, **data)


def dense_k8(P, dx, ds, dc, dd):
    nx, ns, nc, nd = P["nx"], P["ns"], P["nc"], P["nd"]
    nb = [len(P["ixl"]), len(P["ixu"]), len(P["isl"]), len(P["isu"])]
    off = np.cumsum([0, nx, ns, nc, nd] + nb)
    K = np.zeros((off[-1], off[-1]))
    X, S, Cc, Dd, ZL, ZU, VL, VU = [slice(off[i], off[i + 1]) for i in range(8)]
    E = lambda n, idx: np.eye(n)[:, idx]                       # expansion matrix P (n x nb)
    Pxl, Pxu, Pdl, Pdu = E(nx, P["ixl"]), E(nx, P["ixu"]), E(ns, P["isl"]), E(ns, P["isu"])
    K[X, X] = P["W"] + dx * np.eye(nx); K[X, Cc] = P["Jc"].T; K[X, Dd] = P["Jd"].T; K[X, ZL] = -Pxl; K[X, ZU] = Pxu
    K[S, S] = ds * np.eye(ns); K[S, Dd] = -np.eye(ns); K[S, VL] = -Pdl; K[S, VU] = Pdu
    K[Cc, X] = P["Jc"]; K[Cc, Cc] = -dc * np.eye(nc)
    K[Dd, X] = P["Jd"]; K[Dd, S] = -np.eye(nd); K[Dd, Dd] = -dd * np.eye(nd)
    K[ZL, X] = np.diag(P["zl"]) @ Pxl.T; K[ZL, ZL] = np.diag(P["sxl"])
    K[ZU, X] = -np.diag(P["zu"]) @ Pxu.T; K[ZU, ZU] = np.diag(P["sxu"])
    K[VL, S] = np.diag(P["vl"]) @ Pdl.T; K[VL, VL] = np.diag(P["ssl"])
    K[VU, S] = -np.diag(P["vu"]) @ Pdu.T; K[VU, VU] = np.diag(P["ssu"])
    return K, off


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_pd_solve_once_and_residual_match_the_dense_eight_block_system(seed):
    P = build(seed)
    nx, ns, nc, nd = P["nx"], P["ns"], P["nc"], P["nd"]
    rng = np.random.default_rng(100 + seed)
    # Sigma = Z / slack summed onto the x / s diagonal: what makes the reduced (augmented) system equivalent to the 8-block one
    sig_x = np.zeros(nx); np.add.at(sig_x, P["ixl"], P["zl"] / P["sxl"]); np.add.at(sig_x, P["ixu"], P["zu"] / P["sxu"])
    sig_s = np.zeros(ns); np.add.at(sig_s, P["isl"], P["vl"] / P["ssl"]); np.add.at(sig_s, P["isu"], P["vu"] / P["ssu"])
    dx, ds, dc, dd = 30.0, 0.5, 1e-3, 2e-3                # W + Sigma + dx I positive definite (rows of W sum to < 30) => inertia (nx+ns, nc+nd)
    wr, wc, wv = P["wt"]; jcr, jcc, jcv = P["jct"]; jdr, jdc, jdv = P["jdt"]
    ar = lambda n, o: np.arange(n) + o
    irn = np.concatenate([wr, ar(nx, 0), ar(ns, nx), jcr + nx + ns, ar(nc, nx + ns), jdr + nx + ns + nc, ar(nd, nx + ns + nc), ar(nd, nx + ns + nc)]) + 1
    jcn = np.concatenate([wc, ar(nx, 0), ar(ns, nx), jcc, ar(nc, nx + ns), jdc, ar(ns, nx), ar(nd, nx + ns + nc)]) + 1
    lens = [len(wv), nx, ns, len(jcv), nc, len(jdv), ns, nd]
    srcs = [wv, sig_x, sig_s, jcv, np.zeros(nc), jdv, np.zeros(ns), np.zeros(nd)]
    scale = np.array([1, 1, 1, 1, 0, 1, 0, 0], dtype=float); shift = np.array([0, dx, ds, 0, -dc, 0, -1, -dd], dtype=float)
    vals0 = np.concatenate([sc * np.asarray(v) + sh for sc, sh, v in zip(scale, shift, srcs)])
    s = ipopt_amd.KKTSolver(scaling=0)
    s.initialize_structure(nx + ns + nc + nd, irn, jcn, vals=vals0)
    s.assembly_define(lens)
    for q, v in enumerate(srcs):
        s.assembly_set(q, v)
    st, neg, zero = s.factor_assembled(scale, shift)
    assert st == 0 and neg == nc + nd
    nb = [len(P["ixl"]), len(P["ixu"]), len(P["isl"]), len(P["isu"])]
    s.pd_define([nx, ns, nc, nd] + nb, P["ixl"], P["ixu"], P["isl"], P["isu"], irn, jcn, [0, 3, 5])
    s.pd_put_data([P["zl"], P["zu"], P["vl"], P["vu"], P["sxl"], P["sxu"], P["ssl"], P["ssu"]])
    K, off = dense_k8(P, dx, ds, dc, dd)
    split = lambda v: [v[off[i]:off[i + 1]] for i in range(8)]
    rhs = rng.standard_normal(off[-1])
    s.pd_put(0, split(rhs))
    # res <- K8^{-1} rhs
    s.pd_solve_once(0, 1, 1.0, 0.0)
    res = np.concatenate(s.pd_get(1))
    ref = np.linalg.solve(K, rhs)
    assert np.abs(res - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
    # residual and norms at a perturbed point
    pert = res + 1e-3 * rng.standard_normal(off[-1])
    s.pd_put(1, split(pert))
    nr = s.pd_residual(0, 1, 2, [dx, ds, dc, dd])
    resid = np.concatenate(s.pd_get(2))
    rref = K @ pert - rhs
    assert np.abs(resid - rref).max() <= 1e-12 * max(1.0, np.abs(K).sum(axis=1).max() * np.abs(pert).max())
    assert np.allclose(nr, [np.abs(rhs).max(), np.abs(pert).max(), np.abs(rref).max()], rtol=1e-12, atol=0)
    # one refinement step: res <- res - K8^{-1} resid brings the perturbed point back
    s.pd_solve_once(2, 1, -1.0, 1.0)
    back = np.concatenate(s.pd_get(1))
    assert np.abs(back - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
    # res <- alpha sol + beta res, general coefficients
    s.pd_put(1, split(pert))
    s.pd_solve_once(0, 1, 0.5, 2.0)
    mix = np.concatenate(s.pd_get(1))
    assert np.abs(mix - (0.5 * ref + 2.0 * pert)).max() <= 1e-9 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("name", ["hs071", "lukvli1_20"])
def test_pd_kernels_reproduce_the_recorded_solves_of_the_reference(name):
    """Golden fixtures from the UNMODIFIED reference (tests/golden/*.pdrec: what PDFullSpaceSolver::Solve was given and returned, every
    call of hs071 / the first 8 of LukVlI1 n = 20): device-side assembly of the recorded W, J_c, J_d, Sigma and perturbations, then
    solve_once + one refinement step on the device must land on the vector the reference returned, and pd_residual on the oracle's."""
    import os
    from oracle import pd_oracle as po
    recs = po.read_pdrec(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".pdrec"))
    checked = 0
    for r in recs:
        if r["beta"] != 0.0 or r["alpha"] == 0.0:
            continue
        nx, ns, nc, nd = r["nx"], r["ns"], r["nc"], r["nd"]
        dx, ds, dc, dd = r["deltas"]
        (wr, wc, wv), (jcr, jcc, jcv), (jdr, jdc, jdv) = r["W"], r["Jc"], r["Jd"]
        ar = lambda n, o: np.arange(n) + o
        # SymTMatrix triplets may sit in either triangle: the KKT triplet of W is (max, min)
        irn = np.concatenate([np.maximum(wr, wc), ar(nx, 0), ar(ns, nx), jcr + nx + ns, ar(nc, nx + ns), jdr + nx + ns + nc, ar(nd, nx + ns + nc), ar(nd, nx + ns + nc)]) + 1
        jcn = np.concatenate([np.minimum(wr, wc), ar(nx, 0), ar(ns, nx), jcc, ar(nc, nx + ns), jdc, ar(ns, nx), ar(nd, nx + ns + nc)]) + 1
        lens = [len(wv), nx, ns, len(jcv), nc, len(jdv), ns, nd]
        srcs = [wv, r["sigma_x"], r["sigma_s"], jcv, np.zeros(nc), jdv, np.zeros(ns), np.zeros(nd)]
        scale = np.array([1, 1, 1, 1, 0, 1, 0, 0], dtype=float); shift = np.array([0, dx, ds, 0, -dc, 0, -1, -dd], dtype=float)
        vals0 = np.concatenate([sc * np.asarray(v) + sh for sc, sh, v in zip(scale, shift, srcs)])
        s = ipopt_amd.KKTSolver()
        s.initialize_structure(nx + ns + nc + nd, irn, jcn, vals=vals0)
        s.assembly_define(lens)
        for q, v in enumerate(srcs):
            s.assembly_set(q, v)
        st, neg, zero = s.factor_assembled(scale, shift)
        assert st == 0 and neg == nc + nd                      # the perturbations recorded are those of an accepted factorisation
        nb = [len(r["ixl"]), len(r["ixu"]), len(r["isl"]), len(r["isu"])]
        s.pd_define([nx, ns, nc, nd] + nb, r["ixl"], r["ixu"], r["isl"], r["isu"], irn, jcn, [0, 3, 5])
        s.pd_put_data([r["zl"], r["zu"], r["vl"], r["vu"], r["sxl"], r["sxu"], r["ssl"], r["ssu"]])
        s.pd_put(0, po.split(r, r["rhs"]))
        s.pd_solve_once(0, 1, 1.0, 0.0)
        nr = s.pd_residual(0, 1, 2, [dx, ds, dc, dd])
        res1 = np.concatenate(s.pd_get(1))
        resid_o, ratio_o = po.residual(r, r["rhs"], res1)
        assert np.abs(np.concatenate(s.pd_get(2)) - resid_o).max() <= 1e-12 * max(1.0, np.abs(r["rhs"]).max() + np.abs(res1).max() * np.abs(po.k8_dense(r)).sum(axis=1).max())
        assert np.allclose(nr[:2], [np.abs(r["rhs"]).max(), np.abs(res1).max()], rtol=1e-13, atol=0)
        s.pd_solve_once(2, 1, -1.0, 1.0)                          # one refinement step, as min_refinement_steps = 1 makes the reference do
        res = np.concatenate(s.pd_get(1))
        want = r["res_out"] / r["alpha"]
        assert np.abs(res - want).max() <= 1e-7 * max(1.0, np.abs(want).max()), (name, np.abs(res - want).max())
        _, ratio = po.residual(r, r["rhs"], res)
        assert ratio <= 1e-9
        s.close()
        checked += 1
    assert checked >= 6


def test_pd_calls_fail_loudly_without_a_workspace():
    s = ipopt_amd.KKTSolver()
    n = 4
    s.initialize_structure(n, np.arange(1, n + 1), np.arange(1, n + 1), vals=np.ones(n))
    with pytest.raises(ipopt_amd.KKTError):
        s.pd_solve_once(0, 1)

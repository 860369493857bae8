"""-m gpu: parity of the HIP path (through the C ABI) with the oracle and with the reference's own
recorded boundary traffic.

Stated fp64 tolerances:
  * inertia (number of negative eigenvalues): EXACT;
  * scaled residual  ||K x - b||_inf / (||K||_inf ||x||_inf + ||b||_inf)  <= 1e-12 after our solve
    (no internal refinement; Ipopt applies its own, IpPDFullSpaceSolver.cpp:256-346);
  * solution agreement with the oracle / the reference's recorded solution:
    ||x - x_ref||_inf <= 1e-7 max(1, ||x_ref||_inf) on the (well-conditioned) fixtures;
  * repeated factor+solve of the same values: bitwise identical (no atomics on fp data).
"""
import glob
import os

import numpy as np
import pytest

import ipopt_amd
from ipopt_amd import kkt
from oracle import kkt_oracle as ko
from tests.support import kktgen

pytestmark = pytest.mark.gpu
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.kktrec")))
RES_TOL = 1e-12


def sres(K, x, b):
    return np.abs(K @ x - b).max() / (abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max() + 1e-300)


def gpu_factor_solve(n, r, c, v, b, check=False, required=0, **opts):
    s = ipopt_amd.KKTSolver(**opts)
    s.initialize_structure(n, r, c, vals=v)
    s.values()[:] = v
    x = np.array(b, dtype=np.float64, copy=True)
    st = s.multi_solve(True, x, check, required)
    return s, st, x


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_replay_reference_boundary_recordings(path):
    """Replays, call for call, what the reference sent through SparseSymLinearSolverInterface::MultiSolve
    and checks status / inertia / solution against what the reference's own backend returned."""
    rec = ko.read_kktrec(path)
    r, c = ko.rec_triplets(rec)
    n = rec["dim"]
    s = ipopt_amd.KKTSolver()
    first = next(call for call in rec["calls"] if call["new_matrix"])
    s.initialize_structure(n, r, c, vals=first["a"])      # lazy analysis with the first values, as the adapter does
    nsolved = 0
    for call in rec["calls"]:
        if call["new_matrix"]:
            s.values()[:] = call["a"]
        x = call["rhs"].copy()
        st = s.multi_solve(bool(call["new_matrix"]), x, bool(call["check"]), call["required_neg"])
        K = kktgen.to_scipy(n, r, c, call["a"])
        if call["new_matrix"]:
            _, oneg, ozero, _ = ko.factor_solve(n, r, c, call["a"], u=0.01)
            assert s.number_of_neg_evals() == oneg, "inertia differs from the oracle"
        # reference status: MKL hides 'too few negatives' (IpPardisoMKLSolverInterface.cpp:555); otherwise identical
        if not (call["check"] and s.number_of_neg_evals() < call["required_neg"]):
            assert st == call["status"], (st, call["status"])
            if call["new_matrix"] and call["status"] in (0, 2):
                assert s.number_of_neg_evals() == call["neg"]
        if st == kkt.SUCCESS:
            for k in range(x.shape[0]):
                assert sres(K, x[k], call["rhs"][k]) <= RES_TOL
                if call["status"] == 0:
                    ref = call["sol"][k]
                    assert np.abs(x[k] - ref).max() <= 1e-7 * max(1.0, np.abs(ref).max())
            nsolved += 1
    assert nsolved > 0


SEEDED = {
    "lukvl_1e3": lambda: kktgen.lukvl_like(1000, seed=11),
    "lukvl_1e4_config2_shape": lambda: kktgen.lukvl_like(10000, seed=12),          # BASELINE configs[1] sizes
    "lukvl_dc1e-8": lambda: kktgen.lukvl_like(2000, seed=13, delta_c=1e-8),
    "grid10x9_d2c1": lambda: kktgen.grid_kkt(10, 9, dof=2, ncon=1, seed=14),
    "grid24x24_d3c2": lambda: kktgen.grid_kkt(24, 24, dof=3, ncon=2, seed=15),       # fronts > 128: blocked path
    "grid64x64_d1c1": lambda: kktgen.grid_kkt(64, 64, dof=1, ncon=1, seed=16),
    "grid40x40_wide_sigma": lambda: kktgen.grid_kkt(40, 40, dof=2, ncon=2, seed=17, sigma_exp=8.0),
}


@pytest.mark.parametrize("case", sorted(SEEDED))
def test_seeded_systems_against_oracle(case):
    n, r, c, v, neg = SEEDED[case]()
    K = kktgen.to_scipy(n, r, c, v)
    rng = np.random.default_rng(1)
    b = np.stack([K @ np.ones(n), rng.standard_normal(n)])
    s, st, x = gpu_factor_solve(n, r, c, v, b, check=True, required=neg)
    xo, oneg, ozero, _ = ko.factor_solve(n, r, c, v, b, u=0.01)
    assert st == kkt.SUCCESS and s.number_of_neg_evals() == neg == oneg and ozero == 0
    for k in range(2):
        assert sres(K, x[k], b[k]) <= RES_TOL
        ro = sres(K, xo[k], b[k])
        assert ro <= 1e-8          # the checker's own accuracy (Sigma spans 1e-8..1e+8 in the wide_sigma case)
        if ro <= 1e-13:            # solutions are only comparable digit-for-digit when the checker itself is converged;
            # the forward error is condition-dependent: Sigma in 1e-8..1e+8 (wide_sigma) puts cond(K) near 1e10
            tol = 1e-4 if "wide_sigma" in case else 1e-6
            assert np.abs(x[k] - xo[k]).max() <= tol * max(1.0, np.abs(xo[k]).max())
    # bitwise reproducibility
    x2 = b.copy(); s.multi_solve(True, x2)
    assert np.array_equal(x, x2)


def test_wrong_inertia_and_singular_statuses():
    n, r, c, v, neg = kktgen.lukvl_like(400, seed=21)
    v2 = v.copy(); v2[r == c] -= 1e3 * (r[r == c] <= 400)     # indefinite (1,1) block => too many negative eigenvalues
    true_neg = int((np.linalg.eigvalsh(kktgen.to_scipy(n, r, c, v2).toarray()) < 0).sum())
    assert true_neg > neg
    s, st, _ = gpu_factor_solve(n, r, c, v2, np.ones(n), check=True, required=neg)
    assert st == kkt.WRONG_INERTIA and s.number_of_neg_evals() == true_neg   # readable after WRONG_INERTIA (pitfall 3)
    _, oneg, _, _ = ko.factor_solve(n, r, c, v2)
    assert oneg == true_neg
    # rank-deficient Jacobian (two identical constraint rows) with delta_c = 0  =>  SINGULAR, like MA27/MA97/MUMPS
    nx, m = 6, 3
    H = (np.arange(nx), np.arange(nx), np.full(nx, 2.0))
    ji = np.array([0, 0, 1, 1, 2, 2]); jj = np.array([0, 1, 0, 1, 3, 4]); jv = np.array([1.0, 2.0, 1.0, 2.0, 1.0, 1.0])
    n2, r2, c2, v2 = kktgen.kkt_from_blocks(H, np.zeros(nx), (ji, jj, jv), np.zeros(m), nx, m)
    s, st, _ = gpu_factor_solve(n2, r2, c2, v2, np.ones(n2))
    _, _, ozero, _ = ko.factor_solve(n2, r2, c2, v2)
    assert st == kkt.SINGULAR and ozero >= 1
    # ... and Ipopt's cure (delta_c > 0, IpPDPerturbationHandler.cpp:467-470) makes it regular with inertia (nx, m, 0)
    n2, r2, c2, v2 = kktgen.kkt_from_blocks(H, np.zeros(nx), (ji, jj, jv), np.full(m, 1e-8), nx, m)
    s, st, x = gpu_factor_solve(n2, r2, c2, v2, np.ones(n2), check=True, required=m)
    assert st == kkt.SUCCESS


def test_edge_cases_tiny_empty_diagonal_multirhs():
    s, st, x = gpu_factor_solve(1, np.array([1], np.int32), np.array([1], np.int32), np.array([-4.0]), np.array([8.0]))
    assert st == 0 and s.number_of_neg_evals() == 1 and x[0] == -2.0
    s = ipopt_amd.KKTSolver(); s.initialize_structure(0, np.zeros(0, np.int32), np.zeros(0, np.int32))
    assert s.multi_solve(True, np.zeros(0)) == 0
    d = np.array([2.0, -3.0, 5.0, -7.0, 11.0])
    s, st, x = gpu_factor_solve(5, np.arange(1, 6), np.arange(1, 6), d, np.stack([d, 2 * d, -d]))
    assert st == 0 and s.number_of_neg_evals() == 2 and np.allclose(x, [[1] * 5, [2] * 5, [-1] * 5], rtol=1e-15)


def test_a_pool_that_does_not_fit_the_device_is_refused_with_the_numbers(monkeypatch):
    """Every contribution block is resident (one pool L | cb, DESIGN.md "Data layout"; MBndryCntrl_3D N = 78 takes 74 GiB of the 288): a structure whose pool
    does not fit is refused at set-up with the sizes in the message, and factor / solve answer MI355X_KKT_FATAL -- no partial set-up, no CPU fallback.
    MI355X_KKT_POOL_LIMIT_GIB (a cap for a device shared by several handles) stands in for a structure of hundreds of GiB."""
    n, r, c, v, neg = kktgen.grid_kkt(60, 50, dof=3, ncon=2, seed=3)
    monkeypatch.setenv("MI355X_KKT_POOL_LIMIT_GIB", "0.01")
    s = ipopt_amd.KKTSolver()
    assert s.initialize_structure(n, r, c, vals=v) == 0             # the analysis itself succeeds (host) and stays queryable
    assert s.info().nnz_l > 0
    with pytest.raises(kkt.KKTError, match="does not fit the device.*GiB needed.*MI355X_KKT_POOL_LIMIT_GIB"):
        s.multi_solve(True, np.ones(n))
    neg_, zero_ = kkt.C.c_int(0), kkt.C.c_int(0)
    assert s.lib.mi355x_kkt_factor(s._h, None, kkt.C.byref(neg_), kkt.C.byref(zero_)) == kkt.FATAL
    monkeypatch.delenv("MI355X_KKT_POOL_LIMIT_GIB")
    s2, st, x = gpu_factor_solve(n, r, c, v, np.ones(n), check=True, required=neg)      # the same structure without the cap
    assert st == 0


@pytest.mark.parametrize("mode", ["measured-choice", "contexts", "one-after-the-other"])
def test_eight_right_hand_sides_in_one_call_equal_eight_single_solves_bitwise(mode, monkeypatch):
    """(round 6: up to three of the columns are in flight at once, each in a solve context of its own -- stream, work vectors, tags, flags, epoch --, the persistent chain
    sweeps of the contexts taking turns; whether that is used is measured once per structure: all three ways must give the bits of single solves)
    nrhs > 1 (what IpLowRankAugSystemSolver.cpp:435-487 and sIPOPT ask of MultiSolve): all columns go up in one batch, their sweeps run back to back with no
    host synchronisation in between, all solutions come down behind the last one -- and every column is bitwise the solution of a single solve, with a
    leading dimension larger than n as well (the C ABI's `ld`)."""
    if mode == "contexts": monkeypatch.setenv("MI355X_KKT_TUNE", "solve_ctx_force=1")
    if mode == "one-after-the-other": monkeypatch.setenv("MI355X_KKT_DISABLE", "solve_ctx")
    n, r, c, v, neg = kktgen.grid_kkt(48, 40, dof=3, ncon=2, seed=23)
    K = kktgen.to_scipy(n, r, c, v)
    rng = np.random.default_rng(5)
    B = np.ascontiguousarray(rng.standard_normal((8, n)))
    s, st, x0 = gpu_factor_solve(n, r, c, v, B[0].copy(), check=True, required=neg)
    assert st == 0
    singles = []
    for q in range(8):
        x = B[q].copy(); assert s.multi_solve(False, x) == 0; singles.append(x)
    X = B.copy(); assert s.multi_solve(False, X) == 0
    assert all(np.array_equal(X[q], singles[q]) for q in range(8))
    assert max(sres(K, X[q], B[q]) for q in range(8)) <= RES_TOL
    # ld > n through the C ABI directly
    ld = n + 37
    W = np.zeros((8, ld)); W[:, :n] = B
    assert s.lib.mi355x_kkt_solve(s._h, 8, W.ctypes.data, ld) == 0
    assert all(np.array_equal(W[q, :n], singles[q]) for q in range(8)) and not W[:, n:].any()


def test_full_size_properties_config_sized():
    """BASELINE.json sizes without the oracle: LukVlE1-shaped KKT with n = 10^6 variables (dim 1 999 998):
    by-construction inertia, residual, linearity of the solve, idempotence."""
    n, r, c, v, neg = kktgen.lukvl_like(1_000_000, seed=41)
    K = kktgen.to_scipy(n, r, c, v)
    rng = np.random.default_rng(2)
    b1, b2 = K @ np.ones(n), rng.standard_normal(n)
    s, st, x1 = gpu_factor_solve(n, r, c, v, b1, check=True, required=neg)
    assert st == 0 and s.number_of_neg_evals() == neg
    assert sres(K, x1, b1) <= RES_TOL and np.abs(x1 - 1).max() <= 1e-6
    x2 = b2.copy(); s.multi_solve(False, x2)
    x12 = b1 + 2.0 * b2; s.multi_solve(False, x12)
    assert np.abs(x12 - (x1 + 2.0 * x2)).max() <= 1e-9 * np.abs(x12).max()
    x1b = b1.copy(); s.multi_solve(True, x1b)
    assert np.array_equal(x1, x1b)


def test_internal_refinement_improves_an_ill_conditioned_solve():
    """refine_steps > 0: fp64 residual + correction solves on the device (the option Ipopt does not need -- it runs its own
    loop -- but a stand-alone caller of the C ABI does)."""
    n, r, c, v, neg = kktgen.grid_kkt(40, 40, dof=2, ncon=2, seed=17, sigma_exp=8.0)
    K = kktgen.to_scipy(n, r, c, v)
    b = K @ np.ones(n)
    _, st0, x0 = gpu_factor_solve(n, r, c, v, b)
    _, st2, x2 = gpu_factor_solve(n, r, c, v, b, refine_steps=2)
    assert st0 == 0 and st2 == 0
    e0, e2 = np.abs(x0 - 1).max(), np.abs(x2 - 1).max()
    assert sres(K, x2, b) <= RES_TOL and e2 <= max(e0, 1e-12)


@pytest.mark.parametrize("opts", [dict(wide_panels=1), dict(tree_merge=1), dict(leaf_cols=32), dict(ordering=1), dict(scaling=0), dict(use_graph=0),
                          dict(solve_group=1), dict(chain_group=1), dict(chain_group=2, solve_group=1)],
                         ids=["wide_panels", "tree_merge", "leaf_cols", "md_ordering", "no_scaling", "no_graph", "solve_group", "no_chain_group", "chain_group2"])
def test_optional_code_paths_stay_exact(opts):
    """every non-default analysis / kernel option must give the same inertia and a converged solve"""
    if opts.get("wide_panels"):
        n, r, c, v, neg = kktgen.grid_kkt(110, 90, dof=3, ncon=2, seed=31)   # separator fronts of ~1 200 rows: 128-column panels kick in
    else:
        n, r, c, v, neg = kktgen.grid_kkt(48, 44, dof=3, ncon=2, seed=23)
    K = kktgen.to_scipy(n, r, c, v)
    b = K @ np.ones(n)
    s, st, x = gpu_factor_solve(n, r, c, v, b, check=True, required=neg, **opts)
    assert st == 0 and s.number_of_neg_evals() == neg
    assert sres(K, x, b) <= RES_TOL
    if opts.get("wide_panels"):
        assert s.info().maxsupernode > 96          # the panel solve of a > 96-column panel does not stage L11 in LDS (budget): that path ran


@pytest.mark.parametrize("part1", ["tiles64", "tiles128"])
def test_lookahead_split_updates_are_exact_and_reproducible(monkeypatch, part1):
    """the two-stream look-ahead of the group-end trailing updates (normally only on very large fronts) forced onto a
    mid-size system: same inertia, converged solve, bitwise identical to the single-stream factorisation -- with part 1 (the first 256
    columns, on the main stream) in 64 x 64 tiles (k_big_schur_p1, the default where few tiles are in the launch) and in the 128 x 128 ones"""
    if part1 == "tiles128": monkeypatch.setenv("MI355X_KKT_DISABLE", "p1_small")
    n, r, c, v, neg = kktgen.grid_kkt(110, 90, dof=3, ncon=2, seed=31)        # fronts up to ~1 200 rows: several split updates
    K = kktgen.to_scipy(n, r, c, v)
    b = K @ np.ones(n)
    monkeypatch.setenv("MI355X_KKT_DISABLE", "lookahead")
    s0, st0, x0 = gpu_factor_solve(n, r, c, v, b, check=True, required=neg)
    monkeypatch.delenv("MI355X_KKT_DISABLE")
    monkeypatch.setenv("MI355X_KKT_TUNE", "la_min_nt=3,la_min_tiles=0")
    s1, st1, x1 = gpu_factor_solve(n, r, c, v, b, check=True, required=neg)
    assert st0 == st1 == kkt.SUCCESS and s1.number_of_neg_evals() == neg
    assert sres(K, x1, b) <= RES_TOL
    assert np.array_equal(x0, x1)
    x2 = b.copy(); s1.multi_solve(True, x2)
    assert np.array_equal(x1, x2)


@pytest.mark.parametrize("n", [1000, 40000])
def test_leaf_chains_equal_the_level_by_level_schedule_bitwise(monkeypatch, n):
    """the bottom levels of a banded KKT system's tree are chains of fronts of order <= 16: one launch per sweep walks them (k_leaf_chain, k_fwd_leafchain,
    k_bwd_leafchain) with the contribution blocks handed on in registers / LDS -- same arithmetic as the per-level kernels: same inertia, bitwise the same solution,
    and the oracle's"""
    nn, r, c, v, neg = kktgen.lukvl_like(n, seed=41)
    K = kktgen.to_scipy(nn, r, c, v)
    b = K @ np.linspace(1.0, 2.0, nn)
    monkeypatch.setenv("MI355X_KKT_DISABLE", "leafchain")
    s0, st0, x0 = gpu_factor_solve(nn, r, c, v, b, check=True, required=neg)
    monkeypatch.delenv("MI355X_KKT_DISABLE")
    s1, st1, x1 = gpu_factor_solve(nn, r, c, v, b, check=True, required=neg)
    assert st0 == st1 == kkt.SUCCESS and s1.number_of_neg_evals() == s0.number_of_neg_evals() == neg
    assert np.array_equal(x0, x1)
    assert sres(K, x1, b) <= RES_TOL
    xo, oneg, ozero, _ = ko.factor_solve(nn, r, c, v, rhs=b, u=1e-8)
    assert oneg == neg and np.abs(x1 - xo).max() <= 1e-7 * max(1.0, np.abs(xo).max())


@pytest.mark.parametrize("keep", [False, True], ids=["scaling-recomputed", "scaling-kept-on-delta-only-retries"])
def test_device_side_assembly_equals_host_assembly_bitwise(keep, monkeypatch):
    """mi355x_kkt_factor_assembled: values = scale * source + shift per segment, formed on the device, must
    give the factorisation of the host-assembled values bit for bit; a delta-only refactorisation uploads nothing.
    Round 6: on such a retry -- nothing uploaded, the segment scales unchanged, only the shifts (delta_x, delta_c) new -- the Ruiz factors of the
    last equilibration are KEPT (MA97 keeps its scaling until asked, IpMa97SolverInterface.cpp:725-771): same inertia, residual at rounding level,
    the solution of the freshly equilibrated factorisation to 1e-9; a changed segment scale (W_factor 1 -> 0) equilibrates afresh and is bitwise
    the host-assembled factorisation again.  MI355X_KKT_DISABLE=keep_scale: every factorisation equilibrates, every trial bitwise."""
    if not keep:
        monkeypatch.setenv("MI355X_KKT_DISABLE", "keep_scale")
    rng = np.random.default_rng(5)
    nx, m = 300, 120
    # K = [[H + Sigma + dx I, J^T], [J, -dc I]] in Ipopt's segment order  W | D_x | J_c | D_c
    hi = np.concatenate([np.arange(nx), np.arange(nx - 1)]); hj = np.concatenate([np.arange(nx), np.arange(1, nx)])
    hv = np.concatenate([4.0 + rng.random(nx), rng.uniform(-1, 1, nx - 1)])
    Sigma = 10.0 ** rng.uniform(-3, 3, nx)
    ji = np.repeat(np.arange(m), 3); jj = (2 * np.arange(m)[:, None] + np.arange(3)[None, :]).ravel(); jv = rng.uniform(-1, 1, 3 * m); jv[1::3] += 2.0
    r = np.concatenate([np.minimum(hi, hj), np.arange(nx), ji + nx, np.arange(m) + nx]).astype(np.int32) + 1
    c = np.concatenate([np.maximum(hi, hj), np.arange(nx), jj, np.arange(m) + nx]).astype(np.int32) + 1
    n = nx + m
    lens = [len(hv), nx, len(jv), m]

    def host_vals(dx, dc, wf=1.0):
        return np.concatenate([wf * hv, Sigma + dx, jv, np.full(m, -dc)])

    b = rng.standard_normal(n)
    s = ipopt_amd.KKTSolver()
    s.initialize_structure(n, r, c, vals=host_vals(0.0, 0.0))
    s.assembly_define(lens)
    s.assembly_set(0, hv); s.assembly_set(1, Sigma); s.assembly_set(2, jv)          # D_c has no source (NULL vector in Ipopt): scale 0
    for dx, dc, wf in ((0.0, 0.0, 1.0), (1e-4, 0.0, 1.0), (1e-2, 1e-8, 1.0), (0.0, 1e-8, 0.0)):
        st, neg, zero = s.factor_assembled([wf, 1.0, 1.0, 0.0], [0.0, dx, 0.0, -dc])     # nothing uploaded between the trials
        x = b.copy(); s.multi_solve(False, x)
        s2 = ipopt_amd.KKTSolver(); s2.initialize_structure(n, r, c, vals=host_vals(0.0, 0.0))
        s2.values()[:] = host_vals(dx, dc, wf)
        x2 = b.copy(); st2 = s2.multi_solve(True, x2)
        assert st == st2 and neg == s2.number_of_neg_evals()
        if st == 0:
            kept = keep and (dx, dc, wf) in ((1e-4, 0.0, 1.0), (1e-2, 1e-8, 1.0))       # trials 2 and 3: only the shifts differ from the trial before
            if kept:
                K = kktgen.to_scipy(n, r, c, host_vals(dx, dc, wf))
                assert not np.array_equal(x, x2)                                           # (the kept factors ARE other factors)
                assert sres(K, x, b) <= RES_TOL and np.abs(x - x2).max() <= 1e-9 * np.abs(x2).max()
            else:
                assert np.array_equal(x, x2)


def test_scaling_modes_and_factor_exchange():
    """run-time scaling modes (none / Ruiz on device / caller's factors) and the exchange of the factors -- the meaning of
    control.scaling and scale[] in the MA97 call protocol (IpMa97SolverInterface.cpp:641-678)"""
    n, r, c, v, neg = kktgen.grid_kkt(20, 18, dof=2, ncon=1, seed=41, sigma_exp=6.0)
    K = kktgen.to_scipy(n, r, c, v)
    b = K @ np.ones(n)
    s1, st1, x1 = gpu_factor_solve(n, r, c, v, b, check=True, required=neg)           # Ruiz (default)
    f = s1.get_scaling()
    assert st1 == 0 and f.shape == (n,) and np.all(f > 0)
    Ks = abs(K).multiply(f[:, None]).multiply(f[None, :]).tocsr()
    assert 0.2 <= Ks.max(axis=1).toarray().min() and Ks.max() <= 1.0 + 1e-12           # equilibrated: row maxima within [0.2, 1]
    s2 = ipopt_amd.KKTSolver(); s2.initialize_structure(n, r, c, vals=v); s2.set_scaling(2, f); s2.values()[:] = v
    x2 = b.copy(); assert s2.multi_solve(True, x2, True, neg) == 0
    assert np.array_equal(x1, x2)                                                        # the caller-held factors reproduce the run bit for bit
    s0 = ipopt_amd.KKTSolver(); s0.initialize_structure(n, r, c, vals=v); s0.set_scaling(0); s0.values()[:] = v
    x0 = b.copy(); assert s0.multi_solve(True, x0, True, neg) == 0
    assert np.all(s0.get_scaling() == 1.0) and sres(K, x0, b) <= 1e-11


def test_standalone_ruiz_scaling_of_a_triplet_matrix():
    """mi355x_kkt_ruiz_scaling (the TSymScalingMethod hook): same sweeps in numpy"""
    import ctypes as C
    n, r, c, v, neg = kktgen.lukvl_like(500, seed=9)
    out = np.zeros(n)
    lib = kkt.load_library()
    assert lib.mi355x_kkt_ruiz_scaling(0, n, len(v), r.ctypes.data, c.ctypes.data, v.ctypes.data, 1, 4, out.ctypes.data) == 0
    s = np.ones(n)
    for _ in range(4):
        w = np.abs(v) * s[r - 1] * s[c - 1]
        mx = np.zeros(n); np.maximum.at(mx, r - 1, w); np.maximum.at(mx, c - 1, w)
        s = np.where(mx > 0, s / np.sqrt(np.where(mx > 0, mx, 1.0)), s)
    assert np.allclose(out, s, rtol=1e-14, atol=0)


def test_matching_scaling_mode():
    """scaling mode 3: maximum-product matching scaling (MC64-style, host) computed from the values of each factorisation and
    applied on the device like caller-supplied factors; the factors handed back equal the stand-alone routine's"""
    n, r, c, v, neg = kktgen.grid_kkt(20, 18, dof=2, ncon=1, seed=41, sigma_exp=8.0)
    K = kktgen.to_scipy(n, r, c, v)
    b = K @ np.ones(n)
    s, st, x = gpu_factor_solve(n, r, c, v, b, check=True, required=neg, scaling=3)
    assert st == 0 and sres(K, x, b) <= RES_TOL
    f = s.get_scaling()
    ref = np.zeros(n)
    assert kkt.load_library().mi355x_kkt_matching_scaling(n, len(v), r.ctypes.data, c.ctypes.data, v.ctypes.data, 1, ref.ctypes.data, None) == 0
    assert np.allclose(f, ref, rtol=1e-12)
    Ks = abs(K).multiply(f[:, None]).multiply(f[None, :]).tocsr()
    assert Ks.max() <= 1.0 + 1e-10 and np.allclose(Ks.max(axis=1).toarray().ravel(), 1.0, rtol=1e-10)


def test_matching_scaling_is_reused_until_quality_is_asked_for():
    """scaling mode 4 (adapter: mi355x_scaling matching): the matching scaling of the FIRST factorisation serves the following ones -- MA97's
    '...-reuse' switches, IpMa97SolverInterface.cpp:725-771 -- until IncreaseQuality, which has it computed afresh (:824-840)"""
    n, r, c, v, neg = kktgen.grid_kkt(20, 18, dof=2, ncon=1, seed=41, sigma_exp=8.0)
    K = kktgen.to_scipy(n, r, c, v)
    b = K @ np.ones(n)
    s, st, x = gpu_factor_solve(n, r, c, v, b, check=True, required=neg, scaling=4)
    f1 = s.get_scaling().copy()
    ref = np.zeros(n)
    lib = kkt.load_library()
    assert lib.mi355x_kkt_matching_scaling(n, len(v), r.ctypes.data, c.ctypes.data, v.ctypes.data, 1, ref.ctypes.data, None) == 0
    assert st == 0 and sres(K, x, b) <= RES_TOL and np.allclose(f1, ref, rtol=1e-12)
    # a second matrix (other values, same structure): factors of the first one are kept
    rng = np.random.default_rng(5)
    v2 = v * (1.0 + 0.3 * rng.random(len(v)))
    K2 = kktgen.to_scipy(n, r, c, v2); b2 = K2 @ np.ones(n)
    s.values()[:] = v2
    x2 = b2.copy()
    assert s.multi_solve(True, x2, True, neg) == 0 and sres(K2, x2, b2) <= RES_TOL
    assert np.array_equal(s.get_scaling(), f1)
    # IncreaseQuality: computed afresh from the matrix now in place
    assert s.increase_quality()
    x3 = b2.copy()
    assert s.multi_solve(False, x3, True, neg) == 0 and sres(K2, x3, b2) <= RES_TOL
    ref2 = np.zeros(n)
    assert lib.mi355x_kkt_matching_scaling(n, len(v2), r.ctypes.data, c.ctypes.data, v2.ctypes.data, 1, ref2.ctypes.data, None) == 0
    f3 = s.get_scaling()
    assert np.allclose(f3, ref2, rtol=1e-12) and not np.array_equal(f3, f1)


def _device_matching_checks(n, r, c, v, f, eps=1.0 / 64):
    """what the device auction promises (tests/support/auction_spec.py, tests/test_auction_spec.py): every scaled entry <= 1; dual objective within n eps of the
    exact optimum (the host algorithm's factors): 0 <= 2 (sum log s_exact - sum log s) <= n eps"""
    K = kktgen.to_scipy(n, r, c, v)
    Ks = abs(K).multiply(f[:, None]).multiply(f[None, :]).tocsr()
    assert np.all(f > 0) and np.all(np.isfinite(f)) and Ks.max() <= 1.0 + 1e-12
    ref = np.zeros(n)
    assert kkt.load_library().mi355x_kkt_matching_scaling(n, len(v), r.ctypes.data, c.ctypes.data, v.ctypes.data, 1, ref.ctypes.data, None) == 0
    gap = 2.0 * (np.log(ref).sum() - np.log(f).sum())
    assert -1e-8 * n <= gap <= n * eps + 1e-8 * n, (gap, n * eps)
    return Ks


@pytest.mark.parametrize("case", ["grid_small", "lukvl", "grid", "grid_wide_values"])
def test_device_matching_scaling_mode(case):
    """scaling mode 5: the maximum-product matching scaling computed ON THE DEVICE by a Jacobi auction
    (kernels_match.hip.inc; the job of MC64 behind ma97_scaling mc64 / spral_scaling matching).  The factors are feasible duals of the assignment problem:
    every scaled entry <= 1, the dual objective within n/64 of the exact optimum of the host algorithm; on these families every column is matched."""
    from tests.support import auction_spec
    gen = {"grid_small": lambda: kktgen.grid_kkt(20, 18, dof=2, ncon=1, seed=41, sigma_exp=8.0),      # < 1024 columns free from the start: the one-workgroup kernel only
           "lukvl": lambda: kktgen.lukvl_like(60000, seed=5, sigma_scale=1e3),
           "grid": lambda: kktgen.grid_kkt(110, 90, dof=3, ncon=2, seed=31),
           "grid_wide_values": lambda: kktgen.grid_kkt(60, 50, dof=2, ncon=2, seed=7, sigma_exp=8.0)}[case]
    n, r, c, v, neg = gen()
    K = kktgen.to_scipy(n, r, c, v)
    b = K @ np.ones(n)
    s, st, x = gpu_factor_solve(n, r, c, v, b, check=True, required=neg, scaling=5)
    assert st == 0 and sres(K, x, b) <= RES_TOL
    f = s.get_scaling()
    Ks = _device_matching_checks(n, r, c, v, f)
    info = s.info()
    assert info.matching_unmatched == 0 and info.matching_rounds > 0 and info.matching_ms > 0
    assert Ks.max(axis=1).toarray().min() >= 0.9
    A = abs(K).tocsc(); A.sum_duplicates()
    stats = {}
    spec, un = auction_spec.auction_scaling(n, A.indptr, A.indices, A.data, stats=stats)
    assert un == 0
    # the specification works in the caller's numbering, the device in the permuted one: ties on equal values fall differently (other matchings of the same
    # quality), so the two are compared through what both promise -- their dual objectives lie in the same n eps window below the optimum
    assert abs(2.0 * (np.log(spec).sum() - np.log(f).sum())) <= n / 64.0
    print(f"{case}: n={n} device matching {info.matching_ms:.3f} ms, {info.matching_rounds} rounds (specification: {stats['rounds']})")
    # a second factorisation computes it afresh from the new values (mode 5 = per factorisation), deterministically
    x2 = b.copy(); assert s.multi_solve(True, x2, True, neg) == 0 and np.array_equal(s.get_scaling(), f) and np.array_equal(x2, x)


def test_device_matching_scaling_is_reused_until_quality_is_asked_for():
    """scaling mode 6 (adapter: mi355x_scaling matching-device) = mode 5 computed at the first factorisation and kept until IncreaseQuality -- the reuse
    protocol of mode 4 / MA97's '...-reuse' switches (IpMa97SolverInterface.cpp:725-771,824-840)"""
    n, r, c, v, neg = kktgen.grid_kkt(40, 30, dof=2, ncon=1, seed=41, sigma_exp=8.0)
    K = kktgen.to_scipy(n, r, c, v)
    b = K @ np.ones(n)
    s, st, x = gpu_factor_solve(n, r, c, v, b, check=True, required=neg, scaling=6)
    f1 = s.get_scaling().copy()
    assert st == 0 and sres(K, x, b) <= RES_TOL
    _device_matching_checks(n, r, c, v, f1)
    rng = np.random.default_rng(5)
    v2 = v * (1.0 + 0.3 * rng.random(len(v)))
    K2 = kktgen.to_scipy(n, r, c, v2); b2 = K2 @ np.ones(n)
    s.values()[:] = v2
    x2 = b2.copy()
    assert s.multi_solve(True, x2, True, neg) == 0 and sres(K2, x2, b2) <= RES_TOL
    assert np.array_equal(s.get_scaling(), f1)
    assert s.increase_quality()
    x3 = b2.copy()
    assert s.multi_solve(False, x3, True, neg) == 0 and sres(K2, x3, b2) <= RES_TOL
    f3 = s.get_scaling()
    assert not np.array_equal(f3, f1)
    _device_matching_checks(n, r, c, v2, f3)


@pytest.mark.parametrize("seed", range(8))
def test_device_matching_fuzz_random_patterns(seed):
    """the device auction on random patterns with zero diagonals and (odd seeds) structurally deficient pairs of columns -- the cases of
    tests/test_auction_spec.py::test_auction_fuzz_random_patterns through the C ABI: finite positive factors, no scaled entry above 1, at least as many
    unmatched columns as the pattern's structural deficiency, and the same answer when asked twice"""
    from scipy.sparse.csgraph import maximum_bipartite_matching
    from tests.test_auction_spec import random_symmetric_pattern, column_view
    rng = np.random.default_rng(100 + seed)
    n, r, c, v = random_symmetric_pattern(rng, int(rng.integers(40, 400)), 0.02 + 0.05 * rng.random(), 0.4, singletons=(seed % 2) * int(rng.integers(1, 4)))
    s = ipopt_amd.KKTSolver(scaling=5, delay_rounds=0)
    s.initialize_structure(n, r, c, vals=v)
    s.values()[:] = v
    x = np.ones(n)
    st = s.multi_solve(True, x, False, 0)
    assert st in (kkt.SUCCESS, kkt.SINGULAR)
    f = s.get_scaling().copy()
    info = s.info()
    A = column_view(n, r, c, v); A.eliminate_zeros()
    B = A.multiply(f[:, None]).multiply(f[None, :])
    assert np.all(np.isfinite(f)) and np.all(f > 0) and (B.nnz == 0 or B.max() <= 1.0 + 1e-12)
    deficiency = n - int((maximum_bipartite_matching(A.tocsr(), perm_type="column") >= 0).sum())
    assert info.matching_unmatched >= deficiency
    x = np.ones(n); s.multi_solve(True, x, False, 0)
    assert np.array_equal(s.get_scaling(), f)


def test_device_matching_on_a_structurally_deficient_pattern():
    """a pattern without a perfect matching (two columns whose only entries share a row): the auction ends (a column whose best value has fallen below
    -300 stops bidding), reports the columns it left unmatched, and the factors are finite with every scaled entry <= 1; the factorisation reports SINGULAR as without scaling"""
    n = 4
    r = np.array([1, 2, 3, 3, 4], dtype=np.int32); c = np.array([1, 2, 1, 2, 4], dtype=np.int32)
    v = np.array([0.0, 0.0, 2.0, 5.0, 3.0])
    s = ipopt_amd.KKTSolver(scaling=5)
    s.initialize_structure(n, r, c, vals=v)
    s.values()[:] = v
    x = np.ones(n)
    st = s.multi_solve(True, x, False, 0)
    assert st in (kkt.SUCCESS, kkt.SINGULAR)
    info = s.info()
    assert info.matching_unmatched >= 1
    f = s.get_scaling()
    Ks = abs(kktgen.to_scipy(n, r, c, v)).multiply(f[:, None]).multiply(f[None, :])
    assert np.all(np.isfinite(f)) and np.all(f > 0) and Ks.max() <= 1.0 + 1e-12


def test_sync_free_chain_sweeps_match_the_level_by_level_solves(monkeypatch):
    """the flag-synchronised chain sweeps (one launch per run of pure chain levels) against the launch-per-level solves:
    same solution to rounding (the summation order along the chain differs), deterministic across repetitions"""
    n, r, c, v, neg = kktgen.grid_kkt(110, 90, dof=3, ncon=2, seed=31)
    K = kktgen.to_scipy(n, r, c, v)
    rng = np.random.default_rng(4)
    B = [K @ np.ones(n), rng.standard_normal(n), K @ rng.standard_normal(n)]
    monkeypatch.setenv("MI355X_KKT_DISABLE", "chain_solve")
    s0, st0, _ = gpu_factor_solve(n, r, c, v, B[0], check=True, required=neg)
    X0 = []
    for b in B:
        x = b.copy(); s0.multi_solve(False, x); X0.append(x)
    monkeypatch.delenv("MI355X_KKT_DISABLE")
    s1, st1, _ = gpu_factor_solve(n, r, c, v, B[0], check=True, required=neg)
    assert st0 == st1 == kkt.SUCCESS
    for b, x0 in zip(B, X0):
        x = b.copy(); s1.multi_solve(False, x)
        assert sres(K, x, b) <= RES_TOL
        assert np.abs(x - x0).max() <= 1e-11 * max(1.0, np.abs(x0).max())
        x2 = b.copy(); s1.multi_solve(False, x2)
        assert np.array_equal(x, x2)


@pytest.mark.parametrize("gen", ["grid", "band", "hostile", "uneven"])
def test_data_flow_solve_sweeps_at_every_extent(monkeypatch, gen):
    """the data-flow solve sweeps (k_fwd_chain / k_bwd_chain: tagged messages, poller wavefront, dot workgroups, children pulled through the
    inverse row maps) with the segment reaching down to levels of 1 / 8 / 128 / every number of fronts -- small fronts ride as one-link chains,
    far more workgroups than fit on the chip (in-order dispatch is what keeps the waits deadlock-free) -- against the launch-per-level solves:
    same solution to rounding, bitwise reproducible, several right-hand sides through one factorisation, with and without device refinement."""
    if gen == "grid":
        n, r, c, v, neg = kktgen.grid_kkt(60, 52, dof=3, ncon=2, seed=41)
    elif gen == "band":
        n, r, c, v, neg = kktgen.lukvl_like(30000, seed=7)
    elif gen == "uneven":
        n, r, c, v, neg = kktgen.grid_kkt(150, 9, dof=4, ncon=1, seed=43)         # long thin domain: unbalanced tree, chains of very different lengths
    else:
        n, r, c, v = kktgen.hostile_grid_kkt(24, 20, seed=5); neg = None
    K = kktgen.to_scipy(n, r, c, v)
    rng = np.random.default_rng(9)
    B = [K @ np.ones(n), rng.standard_normal(n), K @ rng.standard_normal(n)]
    for refine in (0, 2):
        monkeypatch.setenv("MI355X_KKT_DISABLE", "chain_solve")
        s0, st0, _ = gpu_factor_solve(n, r, c, v, B[0], refine_steps=refine)
        assert st0 == kkt.SUCCESS
        X0 = []
        for b in B:
            x = b.copy(); s0.multi_solve(False, x); X0.append(x)
        monkeypatch.delenv("MI355X_KKT_DISABLE")
        for maxc in ("1", "8", "128", "1000000"):
            monkeypatch.setenv("MI355X_KKT_TUNE", "chain_solve_maxc=" + maxc)
            s1, st1, _ = gpu_factor_solve(n, r, c, v, B[0], refine_steps=refine)
            assert st1 == kkt.SUCCESS and s1.number_of_neg_evals() == s0.number_of_neg_evals()
            for b, x0 in zip(B, X0):
                x = b.copy(); s1.multi_solve(False, x)
                assert np.abs(x - x0).max() <= 1e-9 * max(1.0, np.abs(x0).max()), (gen, maxc, refine)
                x2 = b.copy(); s1.multi_solve(False, x2)
                assert np.array_equal(x, x2)
            Xm = np.stack(B, axis=0).copy()                   # three right-hand sides in one call
            s1.multi_solve(False, Xm)
            for j, x0 in enumerate(X0):
                assert np.abs(Xm[j] - x0).max() <= 1e-9 * max(1.0, np.abs(x0).max())
        monkeypatch.delenv("MI355X_KKT_TUNE")

def test_short_lived_handles_on_recycled_device_memory():
    """many handles set up, used for a factorisation and two solves, and dropped in one process: each new handle is given device memory (and
    finds LDS contents) the previous ones left behind.  Regression for the intermittent all-NaN first solves of round 3: the forward chain
    sweep multiplied zero-padded inverse rows with LDS words it had not written (0 x NaN); and the zero fills of the set-up run on the
    default stream while the solver's streams are non-blocking.  tools/stress_handles.py is the long form."""
    for it in range(24):
        nn = [100, 400, 2000, 5000][it % 4]
        n, r, c, v, neg = kktgen.lukvl_like(nn, seed=it)
        K = kktgen.to_scipy(n, r, c, v)
        b = K @ np.ones(n)
        s, st, x = gpu_factor_solve(n, r, c, v, b, check=True, required=neg)
        assert st == kkt.SUCCESS and s.number_of_neg_evals() == neg
        x2 = b.copy(); s.multi_solve(False, x2)
        assert np.isfinite(x).all() and np.isfinite(x2).all(), it
        assert np.abs(x - 1.0).max() <= 1e-8 and np.abs(x - x2).max() <= 1e-12, it
        del s


def test_fused_pivot_block_and_panel_solve_is_bitwise_identical(monkeypatch):
    """k_big_diag_trsm (pivot block + panel solve of a front in one flag-synchronised launch, used where a level has few
    fronts) against the two separate launches: the same arithmetic, so the same bits"""
    n, r, c, v, neg = kktgen.grid_kkt(110, 90, dof=3, ncon=2, seed=31)
    K = kktgen.to_scipy(n, r, c, v)
    b = K @ np.ones(n)
    monkeypatch.setenv("MI355X_KKT_DISABLE", "fuse_dt")
    s0, st0, x0 = gpu_factor_solve(n, r, c, v, b, check=True, required=neg)
    monkeypatch.delenv("MI355X_KKT_DISABLE")
    s1, st1, x1 = gpu_factor_solve(n, r, c, v, b, check=True, required=neg)
    assert st0 == st1 == kkt.SUCCESS and s0.info().num_two == s1.info().num_two
    assert np.array_equal(x0, x1)
    for _ in range(3):
        x2 = b.copy(); s1.multi_solve(True, x2)
        assert np.array_equal(x1, x2)


@pytest.mark.parametrize("source", ["mbndry3d_14", "grid40x36"])
def test_recycled_contribution_blocks_give_bitwise_the_same_factorisation(source, monkeypatch):
    """symbolic.cpp step 12a: contribution blocks that only carry a contribution to their parent share space over the level schedule.  Only ADDRESSES
    change: inertia, pivot statistics and every bit of the solution must equal those of the layout with every block resident -- on the 3-D system
    whose plan does reuse space (tests/test_symbolic.py checks that it does) and on a 2-D grid, twice each (the second factorisation finds the
    space holding the first one's dead blocks)."""
    if source == "mbndry3d_14":
        n, r, c, v, neg = kktgen.recorded_kkt(os.path.join(os.path.dirname(__file__), "golden", "mbndry3d_14.kktrec"), which=0)
    else:
        n, r, c, v, neg = kktgen.grid_kkt(40, 36, dof=2, ncon=1, seed=8)
    K = kktgen.to_scipy(n, r, c, v)
    b = np.stack([K @ np.ones(n), np.random.default_rng(5).standard_normal(n)])
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MI355X_KKT_RECYCLE", mode)
        s = ipopt_amd.KKTSolver()
        s.initialize_structure(n, r, c, vals=v)
        assert (int(s.symbolic(27, 5)[0]) > 0) == (mode == "1")
        xs = []
        for rep in range(2):
            s.values()[:] = v
            x = b.copy()
            assert s.multi_solve(True, x, True, neg) == kkt.SUCCESS
            xs.append(x)
        I = s.info()
        out[mode] = (xs, s.number_of_neg_evals(), I.num_two, I.num_small, I.cb_doubles)
        for k in range(2):
            assert sres(K, xs[0][k], b[k]) <= RES_TOL
    assert out["0"][1:4] == out["1"][1:4] and out["1"][4] <= out["0"][4]
    for rep in range(2):
        assert np.array_equal(out["0"][0][rep], out["1"][0][rep])

"""-m gpu: the HIP kernels against the pivoting specification (tests/support/mirror.py, pinned on CPU against the
oracle by tests/test_pivoting_spec.py) and the IncreaseQuality contract of the plug-in interface
(reference IpSparseSymLinearSolverInterface.hpp:220; consumer IpPDFullSpaceSolver.cpp:290-301)."""
import numpy as np
import pytest

import ipopt_amd
from ipopt_amd import kkt
from oracle import kkt_oracle as ko
from tests.support import kktgen, mirror

pytestmark = pytest.mark.gpu


def hip_run(n, r, c, v, b, u, **opts):
    s = ipopt_amd.KKTSolver(pivtol=u, **opts)
    s.initialize_structure(n, r, c, vals=v)
    s.values()[:] = v
    x = np.array(b, dtype=np.float64, copy=True)
    st = s.multi_solve(True, x)
    return s, st, x


@pytest.mark.parametrize("kat,msc", [("threshold_kat", 3), ("forced_pivot_kat", 2)])
def test_hip_pivot_statistics_equal_the_specification(kat, msc):
    n, r, c, v = getattr(kktgen, kat)()
    K = kktgen.to_scipy(n, r, c, v).toarray()
    xt = np.arange(1.0, n + 1.0); b = K @ xt
    seen = []
    for u in (1e-8, 1e-4):
        s, st, x = hip_run(n, r, c, v, b, u, max_sn_cols=msc, delay_rounds=0, **kktgen.KAT_OPTS)      # (static pivoting: the per-front rules against their specification)
        xs, spec = mirror.factor_solve_pivoted(mirror.fetch(s), v, b, u=u, u2=1e-4)
        I = s.info()
        assert st == kkt.SUCCESS
        assert (I.num_neg, I.num_zero, I.num_two, I.num_small, I.u_sensitive) == \
               (spec["num_neg"], spec["num_zero"], spec["num_two"], spec["num_delay"], spec["u_sensitive"]), (u, I, spec)
        xo, oneg, _, _ = ko.factor_solve(n, r, c, v, b, u=u)
        assert I.num_neg == oneg                                       # oracle at the same u
        assert np.abs(x - xo).max() <= 1e-9 and np.abs(x - xs).max() <= 1e-9
        seen.append((I.num_two, I.num_small))
    assert seen[0] != seen[1]                                          # u really changes the factorisation


def test_increase_quality_changes_the_factors_or_says_no():
    # (a) a system with a pivot between 1e-8 and 1e-4 of its column: IncreaseQuality -> true, refactorisation differs
    n, r, c, v = kktgen.threshold_kat()
    K = kktgen.to_scipy(n, r, c, v).toarray()
    xt = np.arange(1.0, n + 1.0); b = K @ xt
    s, st, x0 = hip_run(n, r, c, v, b, 1e-8, max_sn_cols=3, **kktgen.KAT_OPTS)
    assert st == 0 and s.info().num_two == 0 and s.info().u_sensitive == 1
    steps = 0
    while s.increase_quality():
        steps += 1
        s.values()[:] = 0.0                       # the host staging buffer is NOT consulted on a refactor (pitfall 7)
        x = b.copy(); assert s.multi_solve(False, x) == 0
        assert np.abs(x - xt).max() <= 1e-9
    assert steps == 3 and s.pivtol == pytest.approx(1e-4)              # 1e-8 -> 1e-6 -> 3.2e-5 -> 1e-4 (u^0.75, capped)
    assert s.info().num_two == 1                                       # the last refactorisation took the 2x2 pivot
    assert not s.increase_quality()                                    # at the maximum
    # (b) a benign KKT system: no pivot decision depends on u.  By default IncreaseQuality still raises u, as the reference adapters do
    #     (IpMa97SolverInterface.cpp:822-854); with smart_quality it answers false at once and leaves u alone
    n, r, c, v, neg = kktgen.grid_kkt(12, 12, dof=2, ncon=1, seed=31)
    K = kktgen.to_scipy(n, r, c, v)
    s, st, x = hip_run(n, r, c, v, K @ np.ones(n), 1e-8)
    assert st == 0 and s.info().u_sensitive == 0 and s.info().num_small == 0
    assert s.increase_quality() and s.pivtol == pytest.approx(1e-6) and s.info().pivtol == pytest.approx(1e-6)
    x2 = (K @ np.ones(n)).copy()
    assert s.multi_solve(False, x2) == 0 and np.array_equal(x2, x)          # ... and the refactorisation it costs is the identical one
    s, st, x = hip_run(n, r, c, v, K @ np.ones(n), 1e-8, smart_quality=1)
    u0 = s.pivtol
    assert not s.increase_quality() and s.pivtol == u0 and s.info().pivtol == pytest.approx(u0)


def test_new_matrix_false_reuses_the_factorisation():
    n, r, c, v, neg = kktgen.grid_kkt(12, 12, dof=2, ncon=1, seed=31)
    K = kktgen.to_scipy(n, r, c, v)
    s, st, x = hip_run(n, r, c, v, K @ np.ones(n), 1e-8)
    b2 = K @ np.arange(n, dtype=np.float64)
    x2 = b2.copy(); assert s.multi_solve(False, x2) == 0
    assert np.abs(x2 - np.arange(n)).max() <= 1e-8 * n
    # an explicit pivtol change + MultiSolve(new_matrix=false) refactors from the device copy of the values
    s.set_pivtol(1e-2); s._refactor = True
    s.values()[:] = 0.0
    x3 = b2.copy(); assert s.multi_solve(False, x3, True, neg) == 0
    assert np.abs(x3 - np.arange(n)).max() <= 1e-8 * n


def test_nearly_dependent_constraint_rows_give_singular():
    """ADVICE r1 (numeric.hip:361): relative zero-pivot test; Ipopt's cure (delta_c > 0) makes the system regular again."""
    for delta, singular in ((0.0, True), (1e-15, True), (1e-4, False)):
        n, r, c, v = kktgen.nearly_dependent_rows(delta)
        for scaling in (0, 1):
            s, st, x = hip_run(n, r, c, v, np.ones(n), 1e-8, scaling=scaling)
            assert (st == kkt.SINGULAR) == singular, (delta, scaling, st)
            if not singular:
                assert st == kkt.SUCCESS and s.number_of_neg_evals() == 3


@pytest.mark.parametrize("u", [1e-8, 1e-4, 0.01])
def test_seeded_systems_at_several_u(u):
    """the by-construction families at every threshold: inertia exact, converged solve, statistics equal to the specification
    (no equilibration, so that the specification sees the same numbers)."""
    for gen in (lambda: kktgen.lukvl_like(1000, seed=11), lambda: kktgen.grid_kkt(24, 24, dof=3, ncon=2, seed=15)):
        n, r, c, v, neg = gen()
        K = kktgen.to_scipy(n, r, c, v)
        b = K @ np.ones(n)
        s, st, x = hip_run(n, r, c, v, b, u, scaling=0, delay_rounds=0)
        _, spec = mirror.factor_solve_pivoted(mirror.fetch(s), v, b, u=u, u2=1e-4)
        I = s.info()
        assert st == 0 and I.num_neg == neg == spec["num_neg"]
        assert (I.num_two, I.num_small) == (spec["num_two"], spec["num_delay"])
        assert I.num_fast_blocks == spec["num_fast"]                   # pivot blocks accepted on the natural-order a-posteriori path
        res = np.abs(K @ x - b).max() / (abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max())
        assert res <= 1e-12


def test_zero_pivot_list_names_the_dependent_rows():
    """mi355x_kkt_zero_pivots on [[I, J^T], [J, 0]] (the matrix TSymLinearSolver::DetermineDependentRows builds,
    IpTSymLinearSolver.cpp:540-716): rows of J that are combinations of earlier ones come back, nothing else."""
    rng = np.random.default_rng(3)
    ncol, nrow = 30, 12
    J = rng.standard_normal((nrow, ncol)) * (rng.random((nrow, ncol)) < 0.3)
    J[np.arange(nrow), np.arange(nrow)] += 2.0
    J[7] = J[2] - 3.0 * J[5]           # dependent
    J[11] = 2.0 * J[0]                 # dependent
    jr, jc = np.nonzero(J)
    n = ncol + nrow
    r = np.concatenate([jr + ncol, np.arange(n)]).astype(np.int32) + 1
    c = np.concatenate([jc, np.arange(n)]).astype(np.int32) + 1
    v = np.concatenate([J[jr, jc], np.ones(ncol), np.zeros(nrow)])
    s = ipopt_amd.KKTSolver(scaling=0, delay_rounds=0)                 # as the adapter's DetermineDependentRows does
    s.initialize_structure(n, r, c, vals=v)
    s.values()[:] = v
    st = s.multi_solve(True, None)
    _, spec = mirror.factor_solve_pivoted(mirror.fetch(s), v, np.ones(n))
    assert s.info().num_zero == spec["num_zero"]
    z = s.zero_pivots() - 1 - ncol                     # row indices of J
    assert st == kkt.SINGULAR and len(z) == 2
    keep = [i for i in range(nrow) if i not in set(z.tolist())]
    assert np.linalg.matrix_rank(J[keep]) == nrow - 2 == np.linalg.matrix_rank(J)      # what is left has full row rank


@pytest.mark.parametrize("u", [1e-8, 0.01])
def test_hostile_grid_hip_equals_the_specification(u):
    """the hostile system of tests/test_pivoting_spec.py (fronts of up to ~300 rows whose pivot blocks leave the blocked a-posteriori path,
    2x2 pivots, forced pivots, a-posteriori failures in the rows below the pivot blocks): every statistic of the HIP factorisation
    equals the specification's, the inertia is the delaying oracle's, and the solve is as accurate as the specification's."""
    n, r, c, v = kktgen.hostile_grid_kkt(16, 16, seed=3)
    K = kktgen.to_scipy(n, r, c, v)
    xt = np.ones(n); b = K @ xt
    s, st, x = hip_run(n, r, c, v, b, u, scaling=0, pivtolmax=max(u, 1e-4), delay_rounds=0)
    xs, spec = mirror.factor_solve_pivoted(mirror.fetch(s), v, b, u=u, u2=max(u, 1e-4))
    _, oneg, ozero, _ = ko.factor_solve(n, r, c, v, b, u=u)
    I = s.info()
    assert st == kkt.SUCCESS
    assert (I.num_neg, I.num_zero, I.num_two, I.num_small, I.u_sensitive, I.num_fast_blocks) == \
           (spec["num_neg"], spec["num_zero"], spec["num_two"], spec["num_delay"], spec["u_sensitive"], spec["num_fast"]), (I, spec)
    assert I.num_neg == oneg and ozero == 0
    assert spec["num_two"] >= 20 and (u < 0.01 or spec["num_delay"] >= 50)
    assert np.abs(x - xt).max() <= 1e-6 and np.abs(x - xs).max() <= 1e-6


@pytest.mark.parametrize("u", [1e-8, 0.01])
def test_small_fronts_static_path_and_its_fallback_equal_the_specification(u):
    """k_front_dpp16 (four fronts of order <= 16 per wavefront, static order, accepted a posteriori) with the strict kernel behind it on a
    banded system whose small diagonals make it reject a part of the fronts at u = 0.01: statistics equal to the specification, inertia the oracle's."""
    n, r, c, v = kktgen.hostile_band_kkt(2000, frac=0.15, tiny=1e-2, seed=4)
    K = kktgen.to_scipy(n, r, c, v)
    xt = np.ones(n); b = K @ xt
    s, st, x = hip_run(n, r, c, v, b, u, scaling=0, pivtolmax=max(u, 1e-4), delay_rounds=0)
    xs, spec = mirror.factor_solve_pivoted(mirror.fetch(s), v, b, u=u, u2=max(u, 1e-4))
    _, oneg, _, _ = ko.factor_solve(n, r, c, v, b, u=u)
    I = s.info()
    assert st == kkt.SUCCESS
    assert (I.num_neg, I.num_zero, I.num_two, I.num_small, I.u_sensitive) == \
           (spec["num_neg"], spec["num_zero"], spec["num_two"], spec["num_delay"], spec["u_sensitive"]), (I, spec)
    assert I.num_neg == oneg and (spec["num_two"] == 0 if u == 1e-8 else spec["num_two"] >= 20)
    assert np.abs(x - xt).max() <= 1e-9 and np.abs(x - xs).max() <= 1e-9


# ------------------------------------------------------------------------------------------------------
# Delayed pivoting across fronts: factor() moves the columns that failed the threshold tests to their
# parent fronts and refactors.  The specification of the loop is tests/support/mirror.py::factor_solve_delayed, pinned on the
# CPU against the delaying oracle (tests/test_pivoting_spec.py); both sides drive the SAME host code for the structural edit.
# ------------------------------------------------------------------------------------------------------
HOSTILE = {
    "grid_1e-9": lambda: kktgen.hostile_grid_kkt(16, 16, seed=3, tiny=1e-9),
    "band_1e-6": lambda: kktgen.hostile_band_kkt(2000, frac=0.15, tiny=1e-6, seed=4),
    "grid_1e-9_dense": lambda: kktgen.hostile_grid_kkt(16, 16, seed=3, tiny=1e-9, frac=0.6),
}


@pytest.mark.parametrize("u", [1e-8, 0.01])
@pytest.mark.parametrize("case", sorted(HOSTILE))
def test_delayed_pivots_hip_reaches_the_oracles_inertia(case, u):
    """where static pivoting ends in forced or zero pivots (SINGULAR), the product moves the failed columns to their parent fronts and refactors:
    the oracle's inertia, no zero pivot, nothing forced.  On the system whose static factorisation is well behaved (grid_1e-9: forced pivots,
    but no growth that amplifies rounding into different decisions) the HIP loop is held to the specification's EXACT sequence -- same columns
    moved in the same rounds into the same structure, same statistics; on the two whose forced pivots grow by 1e9+ (band, dense grid) the
    sequence depends on the last bits of both sides and only the outcome is compared."""
    n, r, c, v = HOSTILE[case]()
    K = kktgen.to_scipy(n, r, c, v)
    xt = np.ones(n); b = K @ xt
    _, oneg, ozero, _ = ko.factor_solve(n, r, c, v, b, u=u)
    rounds = 16 if case == "grid_1e-9_dense" else 8
    s, st, x = hip_run(n, r, c, v, b, u, scaling=0, pivtolmax=max(u, 1e-4), delay_rounds=rounds)
    I = s.info()
    assert st == kkt.SUCCESS and ozero == 0
    assert (I.num_neg, I.num_zero, I.num_small) == (oneg, 0, 0), I                                 # the oracle's inertia, nothing forced
    assert len(s.failed_pivots()) == 0
    if case != "grid_1e-9" or u == 0.01:
        assert I.num_delayed > 0 and 1 <= I.num_restructures <= rounds
    res = np.abs(K @ x - b).max() / (abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max())
    assert res <= (1e-12 if u == 0.01 else 1e-1), res      # (u = 1e-8 admits multipliers of 1e8 per pivot -- MA27 at ma27_pivtol = 1e-8 likewise: Ipopt's answer is IncreaseQuality)
    if case == "grid_1e-9":
        ref = ipopt_amd.KKTSolver(scaling=0)
        ref.initialize_structure(n, r, c, vals=v)
        xs, spec, edits, moved = mirror.factor_solve_delayed(ref, v, b, u=u, u2=max(u, 1e-4), rounds=rounds)
        assert (I.num_delayed, I.num_restructures) == (moved, edits), (I, moved, edits)          # the same columns moved in the same rounds ...
        assert (I.num_two, I.u_sensitive, I.num_fast_blocks) == (spec["num_two"], spec["u_sensitive"], spec["num_fast"]), (I, spec)
        assert np.array_equal(mirror.fetch(s)["perm"], mirror.fetch(ref)["perm"]) and I.nnz_l == ref.info().nnz_l      # ... into the same structure
        assert np.abs(x - xs).max() <= 1e-6
    # the edited structure stays: the next factorisation of the same matrix needs no further edit, and gives the same answer bit for bit
    x2 = b.copy()
    assert s.multi_solve(True, x2) == kkt.SUCCESS and s.info().num_restructures == I.num_restructures and np.array_equal(x2, x)


def test_delays_off_reproduces_static_pivoting_and_failed_pivots_lists_the_forced_columns():
    n, r, c, v = HOSTILE["grid_1e-9"]()
    K = kktgen.to_scipy(n, r, c, v); b = K @ np.ones(n)
    s, st, x = hip_run(n, r, c, v, b, 0.01, scaling=0, pivtolmax=0.01, delay_rounds=0)
    _, spec = mirror.factor_solve_pivoted(mirror.fetch(s), v, b, u=0.01, u2=0.01)
    I = s.info()
    assert st == kkt.SUCCESS and (I.num_zero, I.num_small, I.num_delayed, I.num_restructures) == (spec["num_zero"], spec["num_delay"], 0, 0) and I.num_small > 50
    sym = mirror.fetch(s)
    assert sorted((sym["perm"][np.array(spec["marks"], dtype=int)] + 1).tolist()) == s.failed_pivots().tolist()       # the marks of the kernels are the specification's
    # the static factorisation of the banded sibling is SINGULAR (the deviation VERDICT r03 names); with the default rounds it is not (test above)
    n, r, c, v = HOSTILE["band_1e-6"]()
    s, st, x = hip_run(n, r, c, v, np.ones(n), 1e-8, scaling=0, delay_rounds=0)
    assert st == kkt.SINGULAR and s.info().num_zero > 100


def test_delayed_pivots_keep_the_callers_buffers_and_the_device_state():
    """a structure edit in the middle of the life of a handle: the pinned values buffer the caller holds stays valid, refactor() works from the
    device copy of the values, the scaling mode survives, and device-side assembly keeps its sources."""
    n, r, c, v = HOSTILE["band_1e-6"]()
    K = kktgen.to_scipy(n, r, c, v); b = K @ np.ones(n)
    s = ipopt_amd.KKTSolver(pivtol=0.01, pivtolmax=0.01, scaling=1)
    s.initialize_structure(n, r, c, vals=v)
    buf = s.values()
    addr = buf.ctypes.data
    buf[:] = v
    x = b.copy()
    assert s.multi_solve(True, x) == kkt.SUCCESS
    I = s.info()
    assert I.num_restructures >= 1 and I.num_small == 0 and s.values().ctypes.data == addr
    res = np.abs(K @ x - b).max() / (abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max())
    assert res <= 1e-12
    buf[:] = 0.0                                   # a refactorisation reads the device copy, not the staging buffer
    s._refactor = True
    x2 = b.copy()
    assert s.multi_solve(False, x2) == kkt.SUCCESS and np.array_equal(x2, x)
    # device-side assembly on the edited structure: one segment, scale 2 => the solution halves
    s.assembly_define([len(v)])
    s.assembly_set(0, v)
    st, neg, zero = s.factor_assembled([2.0], [0.0])
    x3 = b.copy()
    assert st == kkt.SUCCESS and s.multi_solve(False, x3) == kkt.SUCCESS
    assert np.abs(2.0 * x3 - x).max() <= 1e-9 * max(1.0, np.abs(x).max())


def test_a_rejected_optimistic_run_leaves_nothing_behind_for_the_full_schedule():
    """One handle, three matrices of one structure: a friendly one (the optimistic schedule: leaf chains + the data-flow launch over the runs of
    small-front levels, k_front_df, accept everything), a hostile one (multipliers of 1e5 .. 1e6 against the static path's bound of 1e4, fronts rejected: factor() repeats with the full schedule and keeps it for the
    life of the handle), another friendly one.  Every solve must be as good as a fresh handle's: a front of order 17 .. 32 that k_front_df had
    marked "done by the static-order kernel" was skipped by the strict kernel of the full schedule ever after (found on LukVlE5, iteration 8)."""
    N = 20000
    n, r, c, vh = kktgen.hostile_band_kkt(N, frac=0.15, tiny=1e-5, seed=4)
    _, r1, c1, v1, _ = kktgen.lukvl_like(N, seed=4)
    _, _, _, v2, _ = kktgen.lukvl_like(N, seed=4, sigma_scale=0.5)
    assert np.array_equal(r, r1) and np.array_equal(c, c1)
    s = ipopt_amd.KKTSolver(pivtol=1e-8, scaling=0, delay_rounds=0)
    s.initialize_structure(n, r, c, vals=v1)
    xt = np.ones(n)
    for v in (v1, vh, v2, v1):
        K = kktgen.to_scipy(n, r, c, v)
        b = K @ xt
        s.values()[:] = v
        x = b.copy()
        st = s.multi_solve(True, x)
        if v is vh:      # (without delayed pivots this matrix may be answered "singular": what matters is that the full schedule has run on this handle)
            assert st in (kkt.SUCCESS, kkt.SINGULAR)
            continue
        assert st == kkt.SUCCESS
        f = ipopt_amd.KKTSolver(pivtol=1e-8, scaling=0, delay_rounds=0)
        f.initialize_structure(n, r, c, vals=v); f.values()[:] = v
        xf = b.copy()
        assert f.multi_solve(True, xf) == kkt.SUCCESS
        assert s.number_of_neg_evals() == f.number_of_neg_evals() == N - 2
        assert np.abs(x - xt).max() <= 1e-7 and np.abs(x - xf).max() <= 1e-7, (np.abs(x - xt).max(), np.abs(x - xf).max())


def test_cost_of_a_delayed_pivot_edit_stays_under_its_ceiling():
    """VERDICT r05 item 2 (regression guard, not the target): a delayed-pivot round is a structure edit on the host + device re-set-up + refactorisation.
    Measured on the GPU box (`tools/delay_cost.py`, `profiles/r06_*`): 22-36 factorisations per edit of 100 columns at KKT dimension 10^6, host part ~0.2 s.
    At KKT dimension 2 * 10^5 the whole cycle must stay below 120 factorisations (the first edit of a process has been seen at 56 on the box, later ones at 22-36), the factorisation after one edit within 10 % of the one before (measured: 1-3 %), the factor grows
    by less than 1 %, inertia and residual unchanged -- what the reference's backends get for free inside one call (IpMa97SolverInterface.cpp:747-771, info.num_delay)
    costs us this much, and no more."""
    import time
    import torch
    n, r, c, v, neg = kktgen.grid_kkt(250, 160, dof=3, ncon=2, seed=77, sigma_exp=6.0)
    assert n >= 200000
    K = kktgen.to_scipy(n, r, c, v)
    s = ipopt_amd.KKTSolver(delay_rounds=0)
    s.initialize_structure(n, r, c, vals=v)
    dv = torch.tensor(v, dtype=torch.float64, device="cuda")
    for _ in range(3):
        st = s.factor_device(dv.data_ptr())
    assert st[0] == 0 and st[1] == neg
    tf = min(_factor_ms(s, dv) for _ in range(5))
    nnz0 = s.info().nnz_l
    cols = np.random.default_rng(7).choice(n, size=100, replace=False) + 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    moved = s.delay_columns(cols)
    st = s.factor_device(dv.data_ptr())
    torch.cuda.synchronize(); cycle_ms = 1e3 * (time.perf_counter() - t0)
    assert moved == 100 and st[0] == 0 and st[1] == neg
    tf_after = min(_factor_ms(s, dv) for _ in range(5))
    assert cycle_ms <= 120.0 * tf, (cycle_ms, tf)
    assert tf_after <= 1.10 * tf, (tf_after, tf)      # (round 6: a delayed column waits further up a FULL chain link instead of cutting it -- 4.35 -> 4.41 ms here; 5.94 before)
    assert s.info().nnz_l <= 1.01 * nnz0
    b = K @ np.ones(n); db = torch.tensor(b, dtype=torch.float64, device="cuda"); dx = torch.empty_like(db)
    s.solve_device2(db.data_ptr(), dx.data_ptr())
    x = dx.cpu().numpy()
    assert np.abs(K @ x - b).max() / (abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max()) <= 1e-12


def _factor_ms(s, dv):
    s.factor_device(dv.data_ptr())
    return s.info().time_factor_ms

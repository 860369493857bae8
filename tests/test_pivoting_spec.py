"""-m 'not gpu': the pivot-threshold contract (u = pivtol) as an executable specification.

tests/support/mirror.py restates the pivoting rules of the HIP kernels (ldlt_reg / k_big_trsm in
ipopt_amd/csrc/numeric.hip) in numpy, walking the symbolic structures the C ABI exports.  Here that
specification is pinned against the CPU oracle AT THE SAME u (inertia, solution) and against LAPACK on
dense copies; tests/test_gpu_pivoting.py then holds the HIP kernels to the specification.

What the reference expects of u (SURVEY 8(a) policy table): ma27_pivtol / ma57_pivtol / ma97_u = 1e-8,
raised by IncreaseQuality as u <- u^0.75 up to 1e-4 (IpMa97SolverInterface.cpp:822-854,
IpMa27TSolverInterface.cpp:724-740), consumed at IpPDFullSpaceSolver.cpp:290-301."""
import numpy as np
import pytest

import ipopt_amd
from oracle import kkt_oracle as ko
from tests.support import kktgen, mirror


def spec_run(n, r, c, v, b, u, u2=1e-4, **opts):
    s = ipopt_amd.KKTSolver(**opts)
    s.initialize_structure(n, r, c, vals=v)
    sym = mirror.fetch(s)
    x, st = mirror.factor_solve_pivoted(sym, v, b, u=u, u2=u2)
    return sym, x, st


def test_threshold_kat_u_changes_the_pivot_sequence_not_the_answer():
    n, r, c, v = kktgen.threshold_kat()
    K = kktgen.to_scipy(n, r, c, v).toarray()
    true_neg = int((np.linalg.eigvalsh(K) < 0).sum())
    xt = np.arange(1.0, n + 1.0); b = K @ xt
    out = {}
    for u in (1e-8, 1e-4):
        sym, x, st = spec_run(n, r, c, v, b, u, max_sn_cols=3, **kktgen.KAT_OPTS)
        assert list(sym["colptr"]) == [0, 3, 5] and list(sym["rowptr"]) == [0, 4, 6]      # front 0: 3 fully-summed columns + 1 update row
        xo, oneg, ozero, _ = ko.factor_solve(n, r, c, v, b, u=u)
        assert st["num_neg"] == oneg == true_neg and st["num_zero"] == ozero == 0           # inertia: exact, oracle at the same u
        assert np.abs(x - xo).max() <= 1e-9 and np.abs(x - xt).max() <= 1e-9
        out[u] = (st, np.abs(x - xt).max())
    assert out[1e-8][0]["num_two"] == 0 and out[1e-4][0]["num_two"] == 1                    # u = 1e-4 rejects the tiny 1x1, takes the 2x2
    assert out[1e-8][0]["u_sensitive"] == 1                                                 # ... and the u = 1e-8 run knows a larger u matters
    assert out[1e-4][1] <= out[1e-8][1]                                                     # the stricter threshold is at least as accurate


def test_forced_pivot_is_reported_as_num_delay():
    n, r, c, v = kktgen.forced_pivot_kat()
    K = kktgen.to_scipy(n, r, c, v).toarray()
    b = K @ np.arange(1.0, n + 1.0)
    res = {}
    for u in (1e-8, 1e-4):
        sym, x, st = spec_run(n, r, c, v, b, u, max_sn_cols=2, **kktgen.KAT_OPTS)
        assert list(sym["colptr"]) == [0, 1, 3, 4]                   # the eps column is a front of its own with one update row
        assert sym["perm"][0] == 1                                   # ... and it is variable 2 (0-based 1)
        _, oneg, _, _ = ko.factor_solve(n, r, c, v, b, u=u)
        assert st["num_neg"] == oneg == 2 and st["num_zero"] == 0
        res[u] = st
    assert res[1e-8]["num_delay"] == 0 and res[1e-4]["num_delay"] == 1
    assert res[1e-8]["u_sensitive"] == 1


@pytest.mark.parametrize("seed", range(6))
def test_spec_inertia_and_solution_on_hostile_random_systems(seed):
    """small dense-ish symmetric indefinite matrices with zero diagonal blocks and entries spread over 12 orders of
    magnitude, no equilibration: inertia must equal LAPACK's at every u; the solve must be backward stable."""
    rng = np.random.default_rng(seed)
    n = 40
    A = np.triu(rng.standard_normal((n, n)) * (rng.random((n, n)) < 0.15))
    A = A + A.T
    d = 10.0 ** rng.uniform(-6, 6, n) * rng.choice([-1, 1], n)
    d[rng.random(n) < 0.3] = 0.0
    A[np.arange(n), np.arange(n)] = d
    A[0, :] += 1e-3; A[:, 0] += 1e-3                                   # keep it connected / nonsingular
    r, c = np.nonzero(np.tril(A) != 0)
    v = A[r, c]
    r = (r + 1).astype(np.int32); c = (c + 1).astype(np.int32)
    K = kktgen.to_scipy(n, r, c, v).toarray()
    w = np.linalg.eigvalsh(K)
    if np.abs(w).min() < 1e-9 * np.abs(w).max():
        pytest.skip("numerically singular draw")
    b = K @ np.ones(n)
    for u in (1e-8, 1e-4, 0.01):
        _, x, st = spec_run(n, r, c, v, b, u, scaling=0)
        assert st["num_neg"] == int((w < 0).sum()) and st["num_zero"] == 0
        res = np.abs(K @ x - b).max() / (np.abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max())
        assert res <= 1e-10 * max(1.0, 1e-8 / u)       # growth is bounded by 1/u: the weakest threshold gets the weakest bound


def test_nearly_dependent_constraint_rows_are_singular_not_noise():
    """a pivot that is pure cancellation noise must be reported as a zero pivot (=> SYMSOLVER_SINGULAR =>
    delta_c path of PDPerturbationHandler), not counted into the inertia with a random sign."""
    for delta, singular in ((0.0, True), (1e-15, True), (1e-4, False)):
        n, r, c, v = kktgen.nearly_dependent_rows(delta)
        _, x, st = spec_run(n, r, c, v, np.ones(n), 1e-8)
        assert (st["num_zero"] > 0) == singular, (delta, st)
        if not singular:
            assert st["num_neg"] == 3
        _, _, ozero, _ = ko.factor_solve(n, r, c, v, np.ones(n), small=1e-14 * np.abs(v).max())
        assert (ozero > 0) == singular


def test_benign_kkt_systems_do_not_depend_on_u():
    """on the by-construction KKT families no pivot is anywhere near the threshold: u_sensitive = 0, i.e. IncreaseQuality
    has nothing to offer and must say so (returns false) instead of triggering identical refactorisations."""
    for gen in (lambda: kktgen.lukvl_like(300, seed=3), lambda: kktgen.grid_kkt(10, 9, dof=2, ncon=1, seed=14)):
        n, r, c, v, neg = gen()
        K = kktgen.to_scipy(n, r, c, v)
        _, x, st = spec_run(n, r, c, v, K @ np.ones(n), 1e-8, scaling=0)
        assert st["num_neg"] == neg and st["num_delay"] == 0 and st["u_sensitive"] == 0
        assert np.abs(x - 1).max() <= 1e-8


def test_blocked_a_posteriori_rule_of_the_big_pivot_blocks():
    """The fast path of a big front's pivot block (kernels_fronts.hip.inc ldlt_blocked_static, mirror.ldlt_block_static): natural order, 1x1 pivots,
    accepted a posteriori iff every multiplier of the block is <= 1 / max(u, u2, 0.01) and no pivot is at the zero threshold -- otherwise
    the strict rule runs on the untouched block.  Either way: inertia as constructed, converged solve; a benign system takes the fast
    path on (nearly) all blocks, a block with a tiny leading diagonal is rejected."""
    n, r, c, v, neg = kktgen.grid_kkt(40, 36, dof=3, ncon=2, seed=5)
    K = kktgen.to_scipy(n, r, c, v)
    b = K @ np.ones(n)
    s = ipopt_amd.KKTSolver(scaling=0)
    s.initialize_structure(n, r, c, vals=v)
    sym = mirror.fetch(s)
    nbig = int((sym["cls"] == 3).sum())
    assert nbig > 5
    out = {}
    for fast in (True, False):
        x, st = mirror.factor_solve_pivoted(sym, v, b, u=1e-8, u2=1e-4, fast_blocks=fast)
        assert st["num_neg"] == neg and st["num_zero"] == 0
        assert np.abs(K @ x - b).max() <= 1e-9 * (abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max())
        out[fast] = st
    assert out[False]["num_fast"] == 0 and 0.5 * nbig <= out[True]["num_fast"] <= nbig
    # the rule itself on a dense block: accepted with modest multipliers, rejected when the first pivot is 1e-6 of its column
    rng = np.random.default_rng(3)
    A = rng.standard_normal((48, 48)); A = A + A.T + 40.0 * np.diag(rng.choice([-1.0, 1.0], 48))
    ok = mirror.ldlt_block_static(A, 48, 1e-8, 1e-4)
    assert ok is not None and ok["nneg"] == int((np.linalg.eigvalsh(A) < 0).sum()) and list(ok["ord"]) == list(range(48))
    A[0, 0] = 1e-6
    assert mirror.ldlt_block_static(A, 48, 1e-8, 1e-4) is None                       # multiplier ~1e6 > 100
    assert mirror.ldlt_block_static(A, 48, 1e-8, 1e-4) is None and mirror.ldlt_front(A.copy(), 48, 1e-8, 1e-4)["nneg"] == int((np.linalg.eigvalsh(A) < 0).sum())


@pytest.mark.parametrize("u", [1e-8, 0.01])
def test_hostile_grid_specification_against_the_delaying_oracle(u):
    """kktgen.hostile_grid_kkt: a third of the Hessian diagonal is ~1e-6 against O(1) couplings, delta_c = 0, fronts of up to ~300 rows.
    The oracle DELAYS a candidate that fails the threshold test (oracle/ldlt_oracle.c); the kernels' rule -- restated by mirror.py -- forces
    it where the front has nothing else to offer and counts it (num_delay), and the big fronts' pivot blocks fall back from the blocked
    a-posteriori path to the strict rule.  Pinned here: the inertia is the oracle's (and LAPACK's) at both thresholds, the solution
    stays accurate, and the hostile features really fire.  (Where static pivoting gives up -- tiny = 1e-9 at u = 1e-8: zero pivots where the
    oracle finds none -- is DESIGN.md's 'no delayed pivoting' deviation, not tested as a success.)"""
    n, r, c, v = kktgen.hostile_grid_kkt(16, 16, seed=3)
    K = kktgen.to_scipy(n, r, c, v)
    true_neg = int((np.linalg.eigvalsh(K.toarray()) < 0).sum())
    xt = np.ones(n); b = K @ xt
    sym, x, st = spec_run(n, r, c, v, b, u, u2=max(u, 1e-4), scaling=0)
    xo, oneg, ozero, _ = ko.factor_solve(n, r, c, v, b, u=u)
    assert st["num_neg"] == oneg == true_neg and st["num_zero"] == ozero == 0
    assert np.abs(xo - xt).max() <= 1e-6 and np.abs(x - xt).max() <= 1e-6
    nbig = int((sym["cls"] == 3).sum())
    assert nbig >= 10 and st["num_fast"] < nbig              # some pivot blocks of the big fronts need the strict rule at either u
    assert st["num_two"] >= 20 and st["u_sensitive"] == 1
    if u == 0.01:
        assert st["num_fast"] < nbig / 2                      # ... most of them at the tight threshold
        assert st["num_delay"] >= 50                          # forced pivots + a-posteriori failures below the pivot blocks


@pytest.mark.parametrize("u", [1e-8, 0.01])
def test_band_with_small_diagonals_static_order_first_then_strict(u):
    """kktgen.hostile_band_kkt (fronts of order <= ~20, 15 % of the Hessian diagonal at 1e-2): the static-order path of the small fronts accepts
    everything at u = 1e-8 and must hand a part of the fronts to the strict rule (2x2 pivots) at u = 0.01; either way the inertia is the oracle's."""
    n, r, c, v = kktgen.hostile_band_kkt(2000, frac=0.15, tiny=1e-2, seed=4)
    K = kktgen.to_scipy(n, r, c, v)
    xt = np.ones(n); b = K @ xt
    sym, x, st = spec_run(n, r, c, v, b, u, u2=max(u, 1e-4), scaling=0)
    _, st_strict = mirror.factor_solve_pivoted(sym, v, b, u=u, u2=max(u, 1e-4), fast_blocks=False)
    xo, oneg, ozero, _ = ko.factor_solve(n, r, c, v, b, u=u)
    assert st["num_neg"] == oneg == st_strict["num_neg"] and st["num_zero"] == ozero == 0 and st["num_delay"] == 0
    assert np.abs(x - xt).max() <= 1e-10 and np.abs(xo - xt).max() <= 1e-7
    assert st_strict["num_two"] >= 100                        # what the strict rule alone would do
    assert (st["num_two"] == 0) if u == 1e-8 else (20 <= st["num_two"] < st_strict["num_two"])


# ------------------------------------------------------------------------------------------------------
# Delayed pivoting across fronts (api.cpp delay_and_refactor + symbolic.cpp restructure_delays): the loop of the C library with
# the numpy specification in place of the HIP factorisation -- the structural edit itself is the product's own host code,
# reached through mi355x_kkt_delay_columns, which works without a GPU.
# ------------------------------------------------------------------------------------------------------
HOSTILE = {
    "grid_1e-9": lambda: kktgen.hostile_grid_kkt(16, 16, seed=3, tiny=1e-9),
    "band_1e-6": lambda: kktgen.hostile_band_kkt(2000, frac=0.15, tiny=1e-6, seed=4),
    "grid_1e-9_dense": lambda: kktgen.hostile_grid_kkt(16, 16, seed=3, tiny=1e-9, frac=0.6),
}


@pytest.mark.parametrize("u", [1e-8, 0.01])
@pytest.mark.parametrize("case", sorted(HOSTILE))
def test_delayed_pivots_reach_the_delaying_oracles_inertia(case, u):
    """where static pivoting ends in zero pivots (SINGULAR) or forced pivots, moving the failed columns to their
    parent fronts and refactoring must end with NO forced and NO zero pivot and the inertia of the oracle, which delays
    (oracle/ldlt_oracle.c) -- within the default number of rounds."""
    n, r, c, v = HOSTILE[case]()
    K = kktgen.to_scipy(n, r, c, v)
    xt = np.ones(n); b = K @ xt
    _, oneg, ozero, _ = ko.factor_solve(n, r, c, v, b, u=u)
    s = ipopt_amd.KKTSolver(scaling=0)
    s.initialize_structure(n, r, c, vals=v)
    _, st0 = mirror.factor_solve_pivoted(mirror.fetch(s), v, b, u=u, u2=max(u, 1e-4))
    nl0 = s.info().nnz_l
    x, st, edits, moved = mirror.factor_solve_delayed(s, v, b, u=u, u2=max(u, 1e-4), rounds=8)
    I = s.info()
    assert ozero == 0 and st["num_neg"] == oneg and st["num_zero"] == 0 and st["num_delay"] == 0 and not st["marks"], (st0, st)
    assert (I.num_delayed, I.num_restructures) == (moved, edits)
    if st0["num_delay"] or st0["num_zero"]:
        assert 1 <= edits <= 8 and moved > 0 and I.nnz_l <= 2 * nl0            # the structure grew, it did not explode
    else:
        assert edits == 0 and I.nnz_l == nl0
    if case == "band_1e-6":
        assert st0["num_zero"] > 100                                        # the static result was SINGULAR: the deviation VERDICT r03 names
    # the edited structure is a valid symbolic factorisation (block elimination walks it with all its assertions) ...
    mirror.factor_solve(mirror.fetch(s), v, b)          # (its inertia / solution mean nothing here: it inverts the hostile pivot blocks without pivoting)
    # ... and with every multiplier bounded by 1/u the factorisation is as accurate as that bound allows
    res = np.abs(K @ x - b).max() / (abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max())
    assert res <= (1e-12 if u == 0.01 else 1e-1), res      # (u = 1e-8 admits multipliers of 1e8 per pivot -- MA27 at ma27_pivtol = 1e-8 likewise: Ipopt's answer is IncreaseQuality)


def test_delay_columns_moves_columns_to_the_parent_front():
    """the structural edit on its own: the moved column leaves its supernode, is a fully-summed column of the parent's, the partition
    still covers every column once, a supernode never exceeds max_sn_cols, and a column that moves again climbs two levels."""
    n, r, c, v, neg = kktgen.grid_kkt(14, 14, dof=3, ncon=2, seed=2)
    K = kktgen.to_scipy(n, r, c, v); b = K @ np.ones(n)
    s = ipopt_amd.KKTSolver(scaling=0)
    s.initialize_structure(n, r, c, vals=v)
    sym = mirror.fetch(s)
    leaf = next(f for f in range(sym["info"].num_sn) if sym["level"][f] == 0 and sym["parent"][f] >= 0 and sym["parent"][sym["parent"][f]] >= 0 and sym["parent"][sym["parent"][sym["parent"][f]]] >= 0)
    col = sym["perm"][sym["colptr"][leaf]]                       # caller's numbering (0-based)
    chain = [leaf]
    while sym["parent"][chain[-1]] >= 0:
        chain.append(sym["parent"][chain[-1]])
    first_cols = [sym["perm"][sym["colptr"][f + 1] - 1] for f in chain]      # the last column of every front on the way up: an anchor that does not move
    def front_of(sy, orig):
        ip = np.empty(n, dtype=int); ip[sy["perm"]] = np.arange(n)
        return int(np.searchsorted(sy["colptr"], ip[orig], side="right") - 1)
    assert s.delay_columns([col + 1]) == 1
    s1 = mirror.fetch(s)
    assert front_of(s1, col) == front_of(s1, first_cols[1])                 # one level up
    assert sorted(s1["perm"].tolist()) == list(range(n)) and np.diff(s1["colptr"]).max() <= 64
    assert s.delay_columns([col + 1]) == 1
    s2 = mirror.fetch(s)
    assert front_of(s2, col) == front_of(s2, first_cols[3])                 # a repeat offender climbs two levels
    x, nb = mirror.factor_solve(s2, v, b)
    assert nb == neg and np.abs(x - 1).max() <= 1e-8
    root = int(np.nonzero(s2["parent"] < 0)[0][0])
    assert s.delay_columns([s2["perm"][s2["colptr"][root]] + 1]) == 0          # a root front has nowhere to delay to
    I = s.info()
    assert (I.num_delayed, I.num_restructures) == (2, 2)

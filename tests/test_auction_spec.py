"""CPU: the numpy specification of the DEVICE matching scaling (tests/support/auction_spec.py; scaling modes 5 / 6, kernels_match.hip.inc) against the
host algorithm (mi355x_kkt_matching_scaling, Duff & Koster: the exact optimum) -- what the auction promises whatever its eps:
  * |s_i a_ij s_j| <= 1 on every entry (the duals are feasible by construction),
  * weak duality sandwich: 0 <= 2 (sum log s_exact - sum log s_auction) <= n eps  (the dual objective is within n eps of the optimum),
  * the unsymmetrised scaled matched entries are >= exp(-eps)."""
import ctypes as C

import numpy as np
import pytest

from ipopt_amd import kkt
from tests.support import auction_spec, kktgen

CASES = {
    "lukvl": lambda: kktgen.lukvl_like(3000, seed=2, sigma_scale=1e3),
    "grid": lambda: kktgen.grid_kkt(30, 25, dof=2, ncon=2, seed=4, sigma_exp=8.0),
    "grid3": lambda: kktgen.grid_kkt(24, 20, dof=3, ncon=2, seed=9, sigma_exp=8.0),
    "band_hostile": lambda: kktgen.hostile_band_kkt(2000, seed=3),
}


def column_view(n, r, c, v):
    A = abs(kktgen.to_scipy(n, r, c, v)).tocsc()
    A.sum_duplicates()
    return A


@pytest.mark.parametrize("case", sorted(CASES))
def test_auction_scaling_properties_against_the_exact_matching(case):
    n, r, c, v = CASES[case]()[:4]
    A = column_view(n, r, c, v)
    eps = 1.0 / 64
    st = {}
    s, unmatched = auction_spec.auction_scaling(n, A.indptr, A.indices, A.data, eps_final=eps, stats=st)
    assert unmatched == 0 and np.all(s > 0) and np.all(np.isfinite(s))
    B = A.multiply(s[:, None]).multiply(s[None, :]).tocsr()
    assert B.max() <= 1.0 + 1e-12
    ref = np.zeros(n); un = C.c_int(-1)
    assert kkt.load_library().mi355x_kkt_matching_scaling(n, len(v), r.ctypes.data, c.ctypes.data, v.ctypes.data, 1, ref.ctypes.data, C.byref(un)) == 0 and un.value == 0
    gap = 2.0 * (np.log(ref).sum() - np.log(s).sum())
    assert -1e-8 * n <= gap <= n * eps + 1e-8 * n, (gap, n * eps)
    # every row of the symmetrically scaled matrix keeps an entry near 1 on these families
    assert B.max(axis=1).toarray().min() >= 0.9


def test_auction_on_a_structurally_deficient_pattern_reports_the_columns_it_could_not_match():
    # two columns whose only entries sit in the same row: one of them cannot be matched; the scaling stays finite and <= 1
    n = 4
    r = np.array([1, 2, 3, 3, 4], dtype=np.int32); c = np.array([1, 2, 1, 2, 4], dtype=np.int32)       # rows/cols 1-based, lower triangle; (3,3) is absent
    v = np.array([0.0, 0.0, 2.0, 5.0, 3.0])
    A = column_view(n, r, c, v)
    s, unmatched = auction_spec.auction_scaling(n, A.indptr, A.indices, A.data, max_rounds=200, phase_rounds=50)
    assert unmatched >= 1 and np.all(np.isfinite(s)) and np.all(s > 0)
    B = A.multiply(s[:, None]).multiply(s[None, :])
    assert B.max() <= 1.0 + 1e-12


def random_symmetric_pattern(rng, n, density, zero_diag_frac, singletons=0):
    """random sparse symmetric matrix (1-based lower triplets) with entries over 12 orders of magnitude, some zero diagonals and -- optionally -- pairs of
    columns whose only entry is in the same row (structural deficiency)"""
    import scipy.sparse as sp
    A = sp.random(n, n, density=density, random_state=np.random.RandomState(int(rng.integers(1 << 30))), format="coo")
    r, c = np.maximum(A.row, A.col), np.minimum(A.row, A.col)
    keep = r != c
    r, c = r[keep], c[keep]
    v = 10.0 ** rng.uniform(-6, 6, len(r)) * rng.choice([-1.0, 1.0], len(r))
    d = np.where(rng.random(n) < zero_diag_frac, 0.0, 10.0 ** rng.uniform(-6, 6, n))
    rows = np.concatenate([r, np.arange(n)]); cols = np.concatenate([c, np.arange(n)]); vals = np.concatenate([v, d])
    if singletons:                       # columns s, s + 1 get zero diagonals and ONE off-diagonal each, both in row t: at most one of them can be matched
        for q in range(singletons):
            s0, t = 2 * q, n - 1 - q
            m = ~(((rows == s0) | (cols == s0) | (rows == s0 + 1) | (cols == s0 + 1)))
            rows, cols, vals = rows[m], cols[m], vals[m]
            rows = np.concatenate([rows, [s0, s0 + 1, t, t]]); cols = np.concatenate([cols, [s0, s0 + 1, s0, s0 + 1]]); vals = np.concatenate([vals, [0.0, 0.0, 3.0, 7.0]])
    return n, (rows + 1).astype(np.int32), (cols + 1).astype(np.int32), vals.astype(np.float64)


@pytest.mark.parametrize("seed", range(12))
def test_auction_fuzz_random_patterns(seed):
    """random patterns with zero diagonals (and, on odd seeds, structurally deficient pairs of columns): the factors are finite and positive, no scaled entry
    exceeds 1, and the columns left unmatched are at least the structural deficiency of the pattern"""
    from scipy.sparse.csgraph import maximum_bipartite_matching
    rng = np.random.default_rng(100 + seed)
    n, r, c, v = random_symmetric_pattern(rng, int(rng.integers(40, 400)), 0.02 + 0.05 * rng.random(), 0.4, singletons=(seed % 2) * int(rng.integers(1, 4)))
    A = column_view(n, r, c, v); A.eliminate_zeros()
    s, unmatched = auction_spec.auction_scaling(n, A.indptr, A.indices, A.data, max_rounds=4000, phase_rounds=256)
    assert np.all(np.isfinite(s)) and np.all(s > 0)
    B = A.multiply(s[:, None]).multiply(s[None, :])
    assert B.nnz == 0 or B.max() <= 1.0 + 1e-12
    deficiency = n - int((maximum_bipartite_matching(A.tocsr(), perm_type="column") >= 0).sum())
    assert unmatched >= deficiency
    if deficiency == 0 and seed % 2 == 0:
        assert unmatched <= max(2, n // 50)          # (the round limit may leave a straggler; the bound on the entries does not depend on it)

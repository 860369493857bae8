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

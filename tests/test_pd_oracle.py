"""CPU: the restatement of the reference's 8-block algebra (oracle/pd_oracle.py) against recordings made at the PDSystemSolver
boundary of the UNMODIFIED reference (tests/golden/*.pdrec, `oracle/ref_driver --record-pd`, tests/golden/make_golden.sh)."""
import os

import numpy as np
import pytest

from oracle import pd_oracle as po

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["hs071", "lukvli1_20"])
def test_restated_solve_once_and_residual_reproduce_what_the_reference_returned(name):
    recs = po.read_pdrec(os.path.join(GOLD, name + ".pdrec"))
    assert len(recs) >= 7
    refined = 0
    for r in recs:
        assert r["ok"]
        K = po.k8_dense(r)
        # (1) reduce -> 4-block solve -> expand IS the solution of the 8-block system (with Sigma = Z / slack summed on the diagonals)
        sx = np.zeros(r["nx"]); np.add.at(sx, r["ixl"], r["zl"] / r["sxl"]); np.add.at(sx, r["ixu"], r["zu"] / r["sxu"])
        ss = np.zeros(r["ns"]); np.add.at(ss, r["isl"], r["vl"] / r["ssl"]); np.add.at(ss, r["isu"], r["vu"] / r["ssu"])
        assert np.allclose(sx, r["sigma_x"], rtol=1e-12, atol=0) and np.allclose(ss, r["sigma_s"], rtol=1e-12, atol=0)
        sol = po.solve_once(r, r["rhs"])
        ref = np.linalg.solve(K, r["rhs"])
        assert np.abs(sol - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
        # (2) the block-by-block residual is K8 res - rhs
        resid, ratio = po.residual(r, r["rhs"], sol)
        assert np.abs(resid - (K @ sol - r["rhs"])).max() <= 1e-10 * max(1.0, np.abs(K).sum(axis=1).max() * np.abs(sol).max())
        # (3) what the reference returned: res = alpha K8^{-1} rhs (+ beta res_in), refined to its residual_ratio_max = 1e-10 unless the
        #     caller allowed an inexact solve
        if r["beta"] == 0.0 and r["alpha"] != 0.0:
            back = r["res_out"] / r["alpha"]
            _, rr = po.residual(r, r["rhs"], back)
            if not r["allow_inexact"]:
                assert rr <= 1e-9, (name, rr)
                refined += 1
            assert np.abs(back - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max())
    assert refined >= 5

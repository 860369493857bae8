"""-m gpu: the two headline configurations, pinned against the REFERENCE's own CPU linear-solver path run on the same box
(oracle/_ref/ref_kkt_solve = the unmodified TripletToCSRConverter + PardisoMKLSolverInterface over oneMKL PARDISO,
IpPardisoMKLSolverInterface.cpp:440-715): bench.py's workload `synth_1e6` (BASELINE.json configs[3]) and its CI-sized
sibling `grid_1e5`.  Inertia: exact.  Solution: ||x - x_ref||_inf <= 1e-7 ||x_ref||_inf (both solve K x = K 1)."""
import json
import os
import subprocess

import numpy as np
import pytest

import bench
import ipopt_amd
from tests.support import kktgen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "oracle", "_ref", "ref_kkt_solve")


def reference_solve(tmp_path, n, r, c, v, b, threads):
    f = tmp_path / "sys.kkt"
    with open(f, "wb") as fh:
        fh.write(np.array([n, len(v)], dtype=np.int32).tobytes()); fh.write(r.astype(np.int32).tobytes())
        fh.write(c.astype(np.int32).tobytes()); fh.write(v.astype(np.float64).tobytes()); fh.write(b.astype(np.float64).tobytes())
    env = dict(os.environ, MKL_NUM_THREADS=str(threads), OMP_NUM_THREADS=str(threads), MKL_DYNAMIC="FALSE")
    out = subprocess.run([TOOL, str(f), "1", "1", str(tmp_path / "x.bin")], capture_output=True, text=True, env=env, timeout=1500).stdout
    j = json.loads(out.strip().splitlines()[-1])
    return j, np.fromfile(tmp_path / "x.bin")


@pytest.mark.skipif(not os.path.exists(TOOL), reason="oracle/_ref not built (needs /root/reference at build time)")
@pytest.mark.parametrize("workload", ["grid_1e5", "synth_1e6"])
def test_headline_workload_matches_reference_pardiso(workload, tmp_path):
    n, r, c, v, neg = bench.make_workload(workload)
    K = kktgen.to_scipy(n, r, c, v)
    b = K @ np.ones(n)
    j, xref = reference_solve(tmp_path, n, r, c, v, b, min(32, os.cpu_count() or 1))
    assert j["status"] == 0
    s = ipopt_amd.KKTSolver(device=0)
    s.initialize_structure(n, r, c, vals=v)
    s.values()[:] = v
    x = b.copy()
    st = s.multi_solve(True, x, True, neg)
    I = s.info()
    rel = float(np.abs(x - xref).max() / np.abs(xref).max())
    rec = dict(workload=workload, n=n, num_neg_hip=I.num_neg, num_neg_pardiso=j["num_neg"], by_construction=neg, rel_diff_vs_pardiso=rel,
               err_hip=float(np.abs(x - 1).max()), err_pardiso=float(np.abs(xref - 1).max()), num_delay=I.num_small, num_two=I.num_two)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"headline_{workload}.json"), "w") as fh:
        json.dump(rec, fh)
    assert st == 0 and I.num_neg == j["num_neg"] == neg, rec            # inertia: ours == reference PARDISO == by construction
    assert rel <= 1e-7, rec

"""MKL PARDISO's own statistics (nnz(L), factorisation GFlop with its METIS ordering) for a bench workload: the yardstick for the
quality of our nested-dissection ordering.  CPU only; uses oracle/_ref/ref_kkt_solve (REF_PARDISO_MSGLVL=1)."""
import sys, os, subprocess, tempfile, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, ipopt_amd
from tests.support import kktgen
for wl in sys.argv[1:]:
    n, r, c, v, neg = bench.make_workload(wl)
    K = kktgen.to_scipy(n, r, c, v); b = K @ np.ones(n)
    with tempfile.NamedTemporaryFile(suffix=".kkt", delete=False) as f:
        f.write(np.array([n, len(v)], dtype=np.int32).tobytes()); f.write(r.astype(np.int32).tobytes()); f.write(c.astype(np.int32).tobytes())
        f.write(v.astype(np.float64).tobytes()); f.write(b.astype(np.float64).tobytes()); path = f.name
    env = dict(os.environ, MKL_NUM_THREADS=str(min(16, os.cpu_count())), REF_PARDISO_MSGLVL="1")
    out = subprocess.run([os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "ref_kkt_solve"), path, "2", "1"], capture_output=True, text=True, env=env)
    os.unlink(path)
    keep = [l.strip() for l in (out.stdout + out.stderr).splitlines() if any(k in l.lower() for k in ("non-zeros in l", "gflop", "non-zeros in u", "number of supernodes", "size of largest"))]
    s = ipopt_amd.KKTSolver(device=-1); s.initialize_structure(n, r, c, vals=v); I = s.info()
    print(wl, "| ours: nnz(L) %.3g  factor GFlop %.1f  maxfront %d  levels %d" % (I.nnz_l, I.flops_factor / 1e9, I.maxfront, I.num_levels))
    for l in dict.fromkeys(keep): print("    MKL:", l)

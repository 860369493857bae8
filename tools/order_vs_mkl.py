"""Ordering quality on a RECORDED Ipopt KKT system (oracle/_ref/ref_driver <problem> <N> --record file.kktrec): our analysis
next to MKL PARDISO's own statistics.  CPU only."""
import sys, os, subprocess, tempfile, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import kkt_oracle as ko
import ipopt_amd
rec = ko.read_kktrec(sys.argv[1])
n = rec['dim']; r, c = ko.rec_triplets(rec)
call = [cl for cl in rec['calls'] if cl.get('new_matrix')][min(5, len(rec['calls']) - 1)]
v = np.asarray(call['a'], np.float64); b = np.asarray(call['rhs'][0], np.float64)
opts = eval(sys.argv[2]) if len(sys.argv) > 2 else {}
s = ipopt_amd.KKTSolver(device=-1, **opts); s.initialize_structure(n, np.asarray(r, np.int32), np.asarray(c, np.int32), vals=v); I = s.info()
print("n %d nnz %d | ours: nnz(L) %.4g  GFlop %.3f  levels %d  maxfront %d  analyse %.2fs" % (n, len(v), I.nnz_l, I.flops_factor / 1e9, I.num_levels, I.maxfront, I.time_analyse))
with tempfile.NamedTemporaryFile(suffix=".kkt", delete=False) as f:
    f.write(np.array([n, len(v)], dtype=np.int32).tobytes()); f.write(np.asarray(r, np.int32).tobytes()); f.write(np.asarray(c, np.int32).tobytes())
    f.write(v.tobytes()); f.write(b.tobytes()); path = f.name
env = dict(os.environ, MKL_NUM_THREADS="8", REF_PARDISO_MSGLVL="1")
out = subprocess.run([os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "_ref", "ref_kkt_solve"), path, "2", "1"], capture_output=True, text=True, env=env)
os.unlink(path)
for l in dict.fromkeys(x.strip() for x in (out.stdout + out.stderr).splitlines()):
    if any(k in l.lower() for k in ("non-zeros in l:", "gflop   for", "largest")): print("   MKL:", l)

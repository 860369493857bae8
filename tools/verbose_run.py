"""Development aid: one factor+solve with verbose analysis/scheduling output (workload name, or grid:NX:NY)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ipopt_amd, bench
from tests.support import kktgen
w = sys.argv[1]
if w.startswith("grid:"):
    _, a, b = w.split(":"); n, r, c, v, neg = kktgen.grid_kkt(int(a), int(b), dof=3, ncon=2, seed=31)
else:
    n, r, c, v, neg = bench.make_workload(w)
s = ipopt_amd.KKTSolver(verbose=1); s.initialize_structure(n, r, c, vals=v); s.values()[:] = v
x = np.ones(n); print(s.multi_solve(True, x, True, neg))

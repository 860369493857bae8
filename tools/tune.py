import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ipopt_amd
from tests.support import kktgen
import bench
def run(wl, **opts):
    n, r, c, v, neg = bench.make_workload(wl)
    s = ipopt_amd.KKTSolver(**opts); s.initialize_structure(n, r, c, vals=v); s.values()[:] = v
    b = np.ones(n); tf=[]; ts=[]
    for _ in range(4):
        x = b.copy(); st = s.multi_solve(True, x, True, neg); J = s.info(); tf.append(J.time_factor_ms); ts.append(J.time_solve_ms)
    I = s.info()
    print(f"{wl:12s} {str(opts):60s} st={st} lev={I.num_levels:3d} nsn={I.num_sn:7d} maxfront={I.maxfront:5d} nnzL={I.nnz_l:10d} flops={I.flops_factor:.3g} big={I.num_big_fronts:5d} analyse={I.time_analyse:.2f}s factor_ms={min(tf):8.3f} solve_ms={min(ts):7.3f}", flush=True)
for wl in sys.argv[1:]:
    for opts in [dict(), dict(nd_leaf=48), dict(nd_leaf=32), dict(nd_leaf=16), dict(nd_leaf=16, nemin=16), dict(nd_leaf=8, nemin=4), dict(nd_leaf=32, nemin=16, max_sn_cols=32), dict(max_sn_cols=32), dict(max_sn_cols=96, nemin=16)]:
        run(wl, **opts)

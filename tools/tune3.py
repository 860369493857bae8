import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ipopt_amd, bench
from tests.support import kktgen
def run(wl, **opts):
    n, r, c, v, neg = bench.make_workload(wl)
    K = kktgen.to_scipy(n, r, c, v)
    s = ipopt_amd.KKTSolver(**opts); s.initialize_structure(n, r, c, vals=v); s.values()[:] = v
    b = K @ np.ones(n); tf=[]; ts=[]
    for _ in range(5):
        x = b.copy(); st = s.multi_solve(True, x, True, neg); J = s.info(); tf.append(J.time_factor_ms); ts.append(J.time_solve_ms)
    I = s.info()
    res = np.abs(K @ x - b).max() / (abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max())
    print(f"{wl:12s} {str(opts):50s} st={st} res={res:.1e} lev={I.num_levels:3d} nsn={I.num_sn:7d} maxfront={I.maxfront:5d} flops={I.flops_factor:.3g} factor_ms={min(tf):8.3f} solve_ms={min(ts):7.3f} step={min(tf)+2*min(ts):.3f}", flush=True)
for wl in sys.argv[1:]:
    for opts in [dict(tree_merge=0), dict(tree_merge=1), dict(tree_merge=1, nd_leaf=16), dict(tree_merge=0, nd_leaf=16), dict(tree_merge=1, nd_leaf=8), dict(tree_merge=1, nd_leaf=16, nemin=16)]:
        run(wl, **opts)

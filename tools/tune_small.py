import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ipopt_amd
from tests.support import kktgen
def run(nx, ny, **opts):
    n, r, c, v, neg = kktgen.grid_kkt(nx, ny, dof=2, ncon=1, seed=3)
    K = kktgen.to_scipy(n, r, c, v); b = K @ np.ones(n)
    s = ipopt_amd.KKTSolver(**opts); s.initialize_structure(n, r, c, vals=v); s.values()[:] = v
    tf = []; ts = []
    for _ in range(5):
        x = b.copy(); st = s.multi_solve(True, x, True, neg); J = s.info(); tf.append(J.time_factor_ms); ts.append(J.time_solve_ms)
    I = s.info()
    res = np.abs(K @ x - b).max() / (abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max())
    print(f"grid {nx}x{ny} n={n} {opts} st={st} res={res:.1e} lev={I.num_levels} maxfront={I.maxfront} factor_ms={min(tf):.3f} solve_ms={min(ts):.3f}", flush=True)
for nx, ny in ((100, 100), (200, 200)):
    for o in (dict(), dict(wide_panels=1), dict(max_sn_cols=32), dict(nd_leaf=64), dict(nemin=32)):
        run(nx, ny, **o)

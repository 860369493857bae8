import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ipopt_amd
from tests.support import kktgen
import bench
def run(wl, **opts):
    n, r, c, v, neg = bench.make_workload(wl)
    K = kktgen.to_scipy(n, r, c, v)
    s = ipopt_amd.KKTSolver(**opts); s.initialize_structure(n, r, c, vals=v); s.values()[:] = v
    b = K @ np.ones(n); tf=[]; ts=[]
    for _ in range(4):
        x = b.copy(); st = s.multi_solve(True, x, True, neg); J = s.info(); tf.append(J.time_factor_ms); ts.append(J.time_solve_ms)
    I = s.info()
    res = np.abs(K @ x - b).max() / (abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max())
    pr = s.profile(3)
    top = sorted(((k_, v_[0] / 3) for k_, v_ in pr.items() if v_[1] > 0), key=lambda t: -t[1])[:int(os.environ.get('TOPK', '6'))]
    print(f"{wl:12s} {str(opts):40s} st={st} res={res:.1e} small={I.num_small} lev={I.num_levels:3d} nsn={I.num_sn:7d} maxfront={I.maxfront:5d} flops={I.flops_factor:.3g} factor_ms={min(tf):8.3f} solve_ms={min(ts):7.3f} | " + " ".join(f"{a}={b:.3f}" for a, b in top), flush=True)
for wl in sys.argv[1:]:
    for opts in [dict()] + [eval(o) for o in os.environ.get('OPTS', '').split(';') if o]:
        run(wl, **opts)

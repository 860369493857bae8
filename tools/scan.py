import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ipopt_amd
from tests.support import kktgen
def run(n_, seed, **opts):
    n, r, c, v, neg = kktgen.lukvl_like(n_, seed=seed)
    K = kktgen.to_scipy(n, r, c, v); b = K @ np.ones(n)
    s = ipopt_amd.KKTSolver(**opts); s.initialize_structure(n, r, c, vals=v); s.values()[:] = v
    x = b.copy(); st = s.multi_solve(True, x, True, neg); I = s.info()
    res = np.abs(K @ x - b).max() / (abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max())
    bad = np.argsort(-np.abs(x - 1))[:3]
    print(f"n={n_} seed={seed} opts={opts} st={st} neg={I.num_neg}/{neg} two={I.num_two} small={I.num_small} zero={I.num_zero} res={res:.2e} err={np.abs(x-1).max():.2e} worst={bad.tolist()}", flush=True)
for n_ in (1000, 10000, 100000, 1000000):
    run(n_, 20260923)
run(1000000, 20260923, scaling=0)
run(1000000, 20260923, pivtol=1e-2)
for sd in (1,2,3):
    run(1000000, sd)

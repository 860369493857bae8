"""stress: small LukVl-like systems, many handles; reports NaN solutions and what fixes them in the same process"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ipopt_amd
from tests.support import kktgen
def run(n, r, c, v, b, **env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        s = ipopt_amd.KKTSolver(device=0)
        s.initialize_structure(n, r, c, vals=v)
        s.values()[:] = v
        x = b.copy(); st = s.multi_solve(True, x)
        x2 = b.copy(); s.multi_solve(False, x2)
        return st, s.info(), x, x2
    finally:
        for k, o in old.items():
            if o is None: os.environ.pop(k, None)
            else: os.environ[k] = o
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    nn = [100, 400, 2000, 5000][it % 4]
    n, r, c, v, neg = kktgen.lukvl_like(nn, seed=it)
    K = kktgen.to_scipy(n, r, c, v); b = K @ np.ones(n)
    st, I, x, x2 = run(n, r, c, v, b)
    nanx, nanx2 = int(np.isnan(x).sum()), int(np.isnan(x2).sum())
    err = float(np.nanmax(np.abs(x - 1.0)))
    if nanx or nanx2 or err > 1e-6 or I.num_neg != neg:
        bad += 1
        print(f"BAD it {it} n {n}: status {st} neg {I.num_neg}/{neg} two {I.num_two} zero {I.num_zero} nan {nanx} resolve-nan {nanx2} err {err:.2e} first idx {np.nonzero(np.isnan(x))[0][:6].tolist()}", flush=True)
        for env in ({"MI355X_KKT_DISABLE": "chain_solve"}, {"MI355X_KKT_DISABLE": "fastpiv"}, {}):
            st_, I_, y, y2 = run(n, r, c, v, b, **env)
            print(f"    retry {env}: nan {int(np.isnan(y).sum())} err {float(np.nanmax(np.abs(y - 1.0))):.2e} neg {I_.num_neg}", flush=True)
print("stress done, bad =", bad)

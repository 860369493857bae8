"""Development aid: run a battery of KKT systems through the C ABI on the GPU and print
inertia / residual / timing diagnostics.  (Parity proper lives in tests/ -m gpu.)"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ipopt_amd
from tests.support import kktgen

def run(name, gen, **opts):
    n, r, c, v, neg = gen()
    K = kktgen.to_scipy(n, r, c, v)
    t0 = time.time()
    s = ipopt_amd.KKTSolver(**opts)
    s.initialize_structure(n, r, c, vals=v)
    I = s.info()
    ta = time.time() - t0
    s.values()[:] = v
    b = K @ np.ones(n)
    x = b.copy()
    st = s.multi_solve(True, x, True, neg)
    I2 = s.info()
    res = np.abs(K @ x - b).max() / (np.abs(b).max() + 1e-300)
    err = np.abs(x - 1).max()
    # repeat for timing
    tf, ts = [], []
    for _ in range(3):
        x2 = b.copy(); s.multi_solve(True, x2); J = s.info(); tf.append(J.time_factor_ms); ts.append(J.time_solve_ms)
    same = np.array_equal(x, x2)
    print(f"{name:28s} n={n:8d} st={ipopt_amd.STATUS[st]:13s} neg={I2.num_neg}/{neg} zero={I2.num_zero} two={I2.num_two} small={I2.num_small} "
          f"res={res:.2e} err={err:.2e} bitrepro={same} nsn={I.num_sn} lev={I.num_levels} maxfront={I.maxfront} nnzL={I.nnz_l} "
          f"analyse={ta:.2f}s factor_ms={min(tf):.3f} solve_ms={min(ts):.3f}", flush=True)
    s.close()

if __name__ == "__main__":
    big = "--big" in sys.argv
    run("tiny3", lambda: (3, np.array([1,2,2,3,3,3]), np.array([1,1,2,1,2,3]), np.array([2.,1,3,1,1,0]), 1))
    run("lukvl200", lambda: kktgen.lukvl_like(200, seed=1))
    run("lukvl200 nograph", lambda: kktgen.lukvl_like(200, seed=1), use_graph=0)
    run("lukvl200 noscale", lambda: kktgen.lukvl_like(200, seed=1), scaling=0)
    run("lukvl200 nomatch dc=1e-8", lambda: kktgen.lukvl_like(200, seed=1, delta_c=1e-8), matching=0)
    run("grid12x10", lambda: kktgen.grid_kkt(12, 10, dof=2, ncon=1, seed=2))
    run("grid8x8 d3c2", lambda: kktgen.grid_kkt(8, 8, dof=3, ncon=2, seed=3))
    run("grid24x24 d3c2 (big)", lambda: kktgen.grid_kkt(24, 24, dof=3, ncon=2, seed=15))
    run("grid64x64 d1c1 (big)", lambda: kktgen.grid_kkt(64, 64, dof=1, ncon=1, seed=16))
    run("grid100x100 d2c1 (big)", lambda: kktgen.grid_kkt(100, 100, dof=2, ncon=1, seed=17))
    run("lukvl1e4", lambda: kktgen.lukvl_like(10000, seed=4))
    run("lukvl1e5", lambda: kktgen.lukvl_like(100000, seed=5))
    if big:
        run("lukvl1e6", lambda: kktgen.lukvl_like(1000000, seed=6))
        run("grid300x300 d2c1", lambda: kktgen.grid_kkt(300, 300, dof=2, ncon=1, seed=18))
        run("grid500x400 d3c2 (config4)", lambda: kktgen.grid_kkt(500, 400, dof=3, ncon=2, seed=19))

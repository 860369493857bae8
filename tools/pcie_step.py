"""PCIe-inclusive step time through the HOST-buffer boundary (what Ipopt uses): values written into the pinned staging
buffer, factorisation, two solves with pageable host right-hand sides -- next to bench.py's device-resident number."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ipopt_amd, bench
from tests.support import kktgen
for wl in sys.argv[1:]:
    n, r, c, v, neg = bench.make_workload(wl)
    K = kktgen.to_scipy(n, r, c, v); b = K @ np.ones(n)
    s = ipopt_amd.KKTSolver(); s.initialize_structure(n, r, c, vals=v)
    def step():
        s.values()[:] = v
        x = b.copy(); st = s.multi_solve(True, x, True, neg)
        x2 = b.copy(); s.multi_solve(False, x2)
        return st, x
    for _ in range(3): st, x = step()
    t0 = time.perf_counter(); reps = 10
    for _ in range(reps): step()
    dt = (time.perf_counter() - t0) / reps
    I = s.info()
    res = float(np.abs(K @ x - b).max() / (abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max()))
    print(f"{wl}: host-buffer step {dt*1e3:.2f} ms (device factor {I.time_factor_ms:.2f} + solve {I.time_solve_ms:.2f} x2), "
          f"{(I.flops_factor + 2*I.flops_solve)/dt/1e9:.0f} GFLOP/s, status {st}, residual {res:.1e}")

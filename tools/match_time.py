"""Device time of the matching scaling on a bench workload: scaling mode 5 (device auction) against the stand-alone host algorithm.  usage: tools/match_time.py [workload ...]"""
import os, sys, time
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, ipopt_amd
from tests.support import kktgen
for wl in (sys.argv[1:] or ["lukvle1_1e6", "synth_1e6"]):
    n, r, c, v, neg = bench.make_workload(wl)
    K = kktgen.to_scipy(n, r, c, v)
    b = K @ np.ones(n)
    s = ipopt_amd.KKTSolver(scaling=5, verbose=1)
    s.initialize_structure(n, r, c, vals=v)
    s.values()[:] = v
    for rep in range(3):
        x = b.copy(); t = time.perf_counter(); st = s.multi_solve(True, x, True, neg); dt = time.perf_counter() - t
        I = s.info()
        print(f"{wl}: mode 5 call {rep}: status {st}, wall {dt * 1e3:.2f} ms, matching {I.matching_ms:.3f} ms in {I.matching_rounds} rounds, {I.matching_unmatched} unmatched, "
              f"factor {I.time_factor_ms:.3f} ms, max|x-1| {np.abs(x - 1).max():.2e}", flush=True)
    f = s.get_scaling()
    A = abs(K).multiply(f[:, None]).multiply(f[None, :]).tocsr()
    rm = A.max(axis=1).toarray().ravel()
    print(f"{wl}: scaled max {A.max():.15g}, row maxima min {rm.min():.4f} / 1e-4 quantile {np.quantile(rm, 1e-4):.4f}", flush=True)
    ref = np.zeros(n); t = time.perf_counter()
    ipopt_amd.load_library().mi355x_kkt_matching_scaling(n, len(v), r.ctypes.data, c.ctypes.data, v.ctypes.data, 1, ref.ctypes.data, None)
    print(f"{wl}: host algorithm {time.perf_counter() - t:.3f} s; dual gap 2(sum log s_host - sum log s_dev) = {2 * (np.log(ref).sum() - np.log(f).sum()):.4g} (bound n/64 = {n / 64:.4g})", flush=True)
    s.set_scaling(1)
    x = b.copy(); s.multi_solve(True, x, True, neg)
    print(f"{wl}: mode 1 (Ruiz) factor {s.info().time_factor_ms:.3f} ms", flush=True)
    s.close()

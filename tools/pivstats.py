import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ipopt_amd, bench
for wl in sys.argv[1:]:
    n, r, c, v, neg = bench.make_workload(wl)
    s = ipopt_amd.KKTSolver(); s.initialize_structure(n, r, c, vals=v); s.values()[:] = v
    x = np.ones(n); st = s.multi_solve(True, x, True, neg); I = s.info()
    print(wl, "n", n, "2x2 pivots", I.num_two, "=> columns in 2x2:", 2 * I.num_two / n, "neg", I.num_neg, "small", I.num_small)

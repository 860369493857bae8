import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ipopt_amd, bench
wl = sys.argv[1]
n, r, c, v, neg = bench.make_workload(wl)
t0 = time.time()
s = ipopt_amd.KKTSolver(verbose=2)
try:
    s.initialize_structure(n, r, c, vals=v)
except Exception as e:
    print("ERR", e)
print("analyse wall", time.time() - t0, "info.time_analyse", s.info().time_analyse)

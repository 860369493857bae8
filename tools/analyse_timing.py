"""Development aid: phase timers of the host analysis (verbose=2) for one workload; runs without a GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ipopt_amd, bench
n, r, c, v, neg = bench.make_workload(sys.argv[1])
s = ipopt_amd.KKTSolver(device=-1, verbose=2)
t = time.time(); s.initialize_structure(n, r, c, vals=v); print("initialize_structure wall", round(time.time() - t, 3), "s")

"""Development aid: phase times of the host analysis (and of the device set-up) on this machine.  usage: tools/analysis_phases.py [workload ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, ipopt_amd
for wl in (sys.argv[1:] or ["lukvle1_1e6", "synth_1e6"]):
    n, r, c, v, neg = bench.make_workload(wl)
    for i in range(3):
        s = ipopt_amd.KKTSolver(verbose=2 if i == 2 else 0)
        t = time.perf_counter(); s.initialize_structure(n, r, c, vals=v); dt = time.perf_counter() - t
        print(f"{wl}: initialize_structure {dt:.3f} s (analysis {s.info().time_analyse:.3f} s)", flush=True)
        s.close()

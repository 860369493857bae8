"""Analysis time of a bench workload (host work; run it on the GPU box's host to compare with DESIGN.md): two analyses in one process,
phase laps on stderr (verbose = 2).  usage: python tools/analysis_time.py synth_1e6 [lukvle1_1e6 ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, ipopt_amd
for wl in sys.argv[1:] or ["synth_1e6"]:
    t = time.time(); n, r, c, v, neg = bench.make_workload(wl); print(f"{wl}: generated in {time.time() - t:.2f} s", flush=True)
    for rep in range(2):
        s = ipopt_amd.KKTSolver(verbose=2)
        t = time.time(); s.initialize_structure(n, r, c, vals=v)
        print(f"{wl}: initialize_structure call {rep}: {time.time() - t:.3f} s wall (analysis + device set-up)", flush=True)
        del s

import sys, os, ctypes as C, numpy as np
os.environ["MI355X_KKT_TRACE"] = "clocks"
sys.path.insert(0, '/root/repo')
import ipopt_amd, bench
n, r, c, v, neg = bench.make_workload(sys.argv[1])
s = ipopt_amd.KKTSolver(); s.initialize_structure(n, r, c, vals=v); s.values()[:] = v
for _ in range(3): s.multi_solve(True, np.ones(n))
out = (C.c_ulonglong * 128)(); s.lib.mi355x_kkt_debug_clocks(s._h, out); o = list(out)
t0 = min(x for x in o[96:120] if x)
for l in range(12):
    a, b = o[96 + 2*l], o[97 + 2*l]
    if a: print(f"level -{12-l}: first quad start {(a-t0)/100:.1f} us, end {(b-t0)/100:.1f} us")

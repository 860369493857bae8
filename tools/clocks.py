import sys, os, ctypes as C, numpy as np
os.environ["MI355X_KKT_DEBUG_CLOCKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ipopt_amd, bench
wl = sys.argv[1] if len(sys.argv) > 1 else "grid_1e5"
n, r, c, v, neg = bench.make_workload(wl)
s = ipopt_amd.KKTSolver(use_graph=0); s.initialize_structure(n, r, c, vals=v); s.values()[:] = v
for _ in range(3): s.multi_solve(True, np.ones(n))
out = (C.c_ulonglong * 16)()
s.lib.mi355x_kkt_debug_clocks(s._h, out)
o = list(out)
print("k =", o[15])
for a, b, name in [(0, 1, "ldlt_reg"), (1, 2, "writeback"), (2, 3, "invert")]:
    cyc = o[2*b] - o[2*a]; wall = o[2*b+1] - o[2*a+1]
    print(f"{name:10s} shader cycles={cyc:8d} wall ticks(100MHz)={wall:6d} -> {wall/100:.2f} us, effective clock {cyc/max(wall,1)*100:.0f} MHz, cycles/pivot={cyc/max(o[15],1):.0f}")

km = o[14]
print("front kernel (last level launched, block 0): k,m =", km // 1000, km % 1000)
for a, b, name in [(4, 5, "assemble"), (5, 6, "ldlt_reg")]:
    cyc = o[2*b] - o[2*a]; wall = o[2*b+1] - o[2*a+1]
    print(f"{name:10s} shader cycles={cyc:8d} -> {wall/100:.2f} us, cycles/pivot={cyc/max(km//1000,1):.0f}")

import sys, os, ctypes as C, numpy as np
os.environ["MI355X_KKT_TRACE"] = "clocks"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ipopt_amd, bench
wl = sys.argv[1] if len(sys.argv) > 1 else "grid_1e5"
n, r, c, v, neg = bench.make_workload(wl)
s = ipopt_amd.KKTSolver(use_graph=0); s.initialize_structure(n, r, c, vals=v); s.values()[:] = v
for _ in range(3): s.multi_solve(True, np.ones(n))
out = (C.c_ulonglong * 128)()
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

e = o[16:40]
print("fast pivot block (block 0 of the last launch), shader cycles from entry:")
print("  loaded+asm %d  colmax %d  factor %d" % (e[1] - e[0], e[2] - e[1], e[3] - e[2]))
for b in range(4):
    prev = e[2] if b == 0 else e[9 + 2 * (b - 1)]
    print("  sub-block %d: panel done +%d, after barrier +%d (from previous)" % (b, e[8 + 2 * b] - prev, e[9 + 2 * b] - prev))

g = o[32:64]
t0 = g[0]
print("fused group launch, group 0 of the last launch: us after role 0 started (100 MHz wall clock)")
names = ["start", "F(q-1) seen", "solved", "S flag", "updated", "F start", "F flag", "end"]
for q in range(4):
    row = g[8 * q:8 * q + 8]
    if row[0] == 0: continue
    print("  role %d: " % q + "  ".join("%s %.1f" % (names[i], (row[i] - t0) / 100.0) for i in range(8) if t0 <= row[i] < t0 + 10**8))      # (a stamp that this schedule does not take stays 0 or keeps an old value)

ps = o[40:48]
if all(0 <= ps[i + 1] - ps[i] < 10**8 for i in range(7)):
    print("sub-block 0, wavefront 0 (cycles): load %d  factor %d  checks %d  inverse %d  store %d | to barrier B %d, to barrier C %d" % (ps[1]-ps[0], ps[2]-ps[1], ps[3]-ps[2], ps[4]-ps[3], ps[5]-ps[4], ps[6]-ps[5], ps[7]-ps[6]))

r = o[64:92]
if r[4] and r[0]:
    base = r[4]
    names = {4: "flag seen", 0: "L11/D/Is loaded", 1: "rows permuted (+inverses)", 2: "substitution done", 3: "W/L stored", 5: "S flag raised", 6: "other links' columns updated", 7: "own block updated",
             8: "pivot block: entered colmax", 9: "colmax done", 10: "blocked LDL^T done", 11: "L11 / D / Is written", 12: "F flag raised"}
    print("role 1 of group 0 (last launch), last link: shader cycles after the pivot-block flag was seen")
    names.update({13: "own block: tile decoded", 14: "own block: products done", 15: "own block: written"})
    for i in (4, 0, 1, 2, 3, 5, 6, 13, 14, 15, 7, 8, 9, 10, 11, 12):
        print("  %-34s %8d" % (names[i], r[i] - base))

if r[4] and not r[0] and r[8]:      # optimistic schedule: the panel solve ran in step with the pivot block before (trsm_rows_pipe) -- what is left behind its LAST sub-block
    base = r[4]
    print("role 1 of group 0 (last launch), last link, panel solve in step with the pivot block: shader cycles after the LAST 16-column sub-block of the pivot block before was seen")
    for i, nm in ((4, "last sub-block seen"), (2, "last step solved, scaled, stored"), (7, "own block updated (+ S flag)"), (8, "pivot block: entered colmax"), (9, "colmax done"), (10, "blocked LDL^T done"),
                  (11, "L11 / D / Is written"), (12, "F flag raised")):
        print("  %-34s %8d" % (nm, r[i] - base))
if r[4] and r[16]:
    print("  blocked LDL^T of role 1's pivot block, per 16-column sub-block (cycles): diagonal block + inverse | rows below | trailing tiles")
    prev = r[9]
    for b in range(4):
        a_, b_, c_ = r[16 + 3 * b], r[17 + 3 * b] if b < 3 else 0, r[18 + 3 * b] if b < 3 else 0
        print("    sub-block %d: %6d | %6d | %6d" % (b, a_ - prev, (b_ - a_) if b_ else 0, (c_ - b_) if c_ else 0))
        prev = c_ if c_ else a_

#!/usr/bin/env python3
"""Per-level work model of a bench workload (host only): fronts, sizes, flops of the pivot blocks / panel solves / trailing updates and
the bytes of the contribution blocks, level by level -- to be read next to tools/timeline.sh's launch list (which launches are far from
what their work would take at the measured MFMA / HBM rates).   python tools/level_model.py [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, ipopt_amd

wl = sys.argv[1] if len(sys.argv) > 1 else "synth_1e6"
n, r, c, v, _ = bench.make_workload(wl)
s = ipopt_amd.KKTSolver(device=-1)
s.initialize_structure(n, r, c, vals=v)
I = s.info(); g = s.symbolic
colptr, rowptr = g(1, I.num_sn + 1).astype(np.int64), g(2, I.num_sn + 1).astype(np.int64)
lev = g(5, I.num_sn); parent = g(4, I.num_sn)
gpos, grem = g(15, I.num_sn), g(16, I.num_sn)
alias = g(17, I.num_sn)
k = np.diff(colptr); m = np.diff(rowptr); mu = m - k
big = m > 128
print(f"{wl}: n={n} nsn={I.num_sn} levels={lev.max()+1} big fronts={big.sum()}")
print(f"{'lev':>3} {'fronts':>7} {'big':>6} {'m min/med/max':>16} {'k med':>5} {'GF diag':>8} {'GF trsm':>8} {'GF schur':>9} {'MB cb':>8} {'MB L':>8} {'alias':>6}")
for L in range(lev.max() + 1):
    q = (lev == L)
    qb = q & big
    if not qb.any():
        if q.any(): print(f"{L:3d} {q.sum():7d} {0:6d} {m[q].min():5d}/{int(np.median(m[q])):5d}/{m[q].max():5d}")
        continue
    mm, kk, uu = m[qb].astype(float), k[qb].astype(float), mu[qb].astype(float)
    fd = (kk ** 3 / 3).sum() / 1e9; ft = (uu * kk * kk).sum() / 1e9; fs = (uu * uu * kk).sum() / 1e9
    cb = (uu * uu * 4).sum() / 1e6; lb = (mm * kk * 8).sum() / 1e6
    print(f"{L:3d} {q.sum():7d} {qb.sum():6d} {int(mm.min()):5d}/{int(np.median(mm)):5d}/{int(mm.max()):5d} {int(np.median(kk)):5d} {fd:8.2f} {ft:8.2f} {fs:9.2f} {cb:8.1f} {lb:8.1f} {(alias[qb] >= 0).sum():6d}")

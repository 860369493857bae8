#!/usr/bin/env python3
"""What a RECYCLING plan for the contribution blocks would need (host only; the product keeps every block resident -- DESIGN.md "Data layout").
A block of a front that is not in place on a child holds the whole in-place chain above it; it is written from the tree level of that front on and
dead once the parent of the chain's LAST link has been formed -- plus a window of levels for what may still run next to it (look-ahead streams,
chain groups factored at the level of their first link, the data-flow launches that span several levels).  Peak of the live blocks over the level
schedule = the pool a static plan (interval colouring over levels) could get by with.   usage: python tools/mem_plan.py <workload | npz:path> [window ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, ipopt_amd

wl = sys.argv[1] if len(sys.argv) > 1 else "synth_1e6"
windows = [int(a) for a in sys.argv[2:]] or [0, 4, 8, 16]
n, r, c, v, _ = bench.make_workload(wl)
s = ipopt_amd.KKTSolver(device=-1)
s.initialize_structure(n, r, c, vals=v)
I = s.info(); g = s.symbolic; N = I.num_sn
colptr, rowptr = g(1, N + 1).astype(np.int64), g(2, N + 1).astype(np.int64)
parent, lev, alias = g(4, N), g(5, N), g(17, N)
k = np.diff(colptr); m = np.diff(rowptr); mu = m - k
top = np.arange(N)                       # last link of the in-place chain that starts at a front
for sn in range(N):                      # children precede parents in a chain: alias[sn] < sn
    if alias[sn] >= 0: top[alias[sn]] = sn
for sn in range(N - 1, -1, -1):
    if top[sn] != sn: top[sn] = top[top[sn]]
own = alias < 0                          # fronts that own a block
size = (mu.astype(np.float64) ** 2) * 8.0
born = lev.copy()
ptop = parent[top]
dies = np.where(ptop >= 0, lev[np.maximum(ptop, 0)], lev.max())      # the level at which the chain's last block is consumed
L = int(lev.max()) + 1
host = own & (top != np.arange(N))      # the block holds the panels of the in-place links above its front: part of the FACTOR, it stays
total = size[own].sum()
print(f"{wl}: n={n} fronts={N} levels={L}; every block resident: {total / 2**30:.2f} GiB of contribution blocks (+ {8.0 * (m * k)[own].sum() / 2**30:.2f} GiB of panels)")
free = own & ~host
print(f"  blocks that host an in-place chain (they hold that chain's panels: factor storage, resident): {size[host].sum() / 2**30:.2f} GiB in {host.sum()} blocks; "
      f"blocks that only carry a contribution to the parent: {size[free].sum() / 2**30:.2f} GiB in {free.sum()} blocks")
for w in windows:
    live = np.zeros(L + 1)
    b = born[free]; d = np.minimum(dies[free] + w, L - 1)
    np.add.at(live, b, size[free]); np.add.at(live, d + 1, -size[free])
    peak = np.cumsum(live)[:L].max()
    print(f"  window of {w:2d} levels behind the consuming level: peak of the live recyclable blocks {peak / 2**30:7.2f} GiB -> pool {(peak + size[host].sum()) / 2**30:7.2f} GiB = "
          f"{100 * (peak + size[host].sum()) / total:5.1f} % of today's")

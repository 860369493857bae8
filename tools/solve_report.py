"""Reads a rocprofv3 kernel_trace.csv; reports the launches of the LAST triangular solve (k_load_rhs ... k_store_sol)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["k"] = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mi355x::", "")
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if r["k"].startswith("k_load_rhs")]
ends = [i for i, r in enumerate(rows) if r["k"].startswith("k_store_sol")]
i0 = starts[-1]; i1 = [e for e in ends if e > i0][0]
S = rows[i0:i1 + 1]
t0 = S[0]["s"]
print(f"last solve: {len(S)} launches, wall {(S[-1]['e'] - t0) / 1e6:.3f} ms")
by = collections.defaultdict(lambda: [0.0, 0])
for r in S:
    by[r["k"]][0] += (r["e"] - r["s"]) / 1e3; by[r["k"]][1] += 1
for k, (t, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"  {k[:40]:40s} n={n:5d} total={t / 1e3:8.3f} ms mean={t / n:8.1f} us")
for r in S:
    if "chain" in r["k"]:
        print(f"    {(r['s'] - t0) / 1e3:9.1f} us  {(r['e'] - r['s']) / 1e3:8.1f} us  {r['k']:14s} grid {r.get('Grid_Size_X', r.get('Grid_Size', ''))}")
if len(sys.argv) > 2:      # every launch of the last solve, one line each
    with open(sys.argv[2], "w") as f:
        for r in S:
            f.write(f"{(r['s'] - t0) / 1e3:10.1f} {(r['e'] - r['s']) / 1e3:8.1f} {r['k'][:28]:28s} {r.get('Grid_Size_X', '')}x{r.get('Grid_Size_Y', '')} wg {r.get('Workgroup_Size_X', '')}\n")

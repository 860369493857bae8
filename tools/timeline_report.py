"""Reads a rocprofv3 kernel_trace.csv; reports on the LAST factorisation (delimited by k_gather_values ... k_reduce_stats)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["k"] = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mi355x::", "")
    r["q"] = r.get("Queue_Id", r.get("Stream_Id", "?"))
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if r["k"].startswith("k_gather_values")]
ends = [i for i, r in enumerate(rows) if r["k"].startswith("k_reduce_stats")]
i0 = starts[-1]; i1 = [e for e in ends if e > i0][0]
F = rows[i0:i1 + 1]
t0 = F[0]["s"]; t1 = max(r["e"] for r in F)
print(f"last factorisation: {len(F)} launches, wall {(t1 - t0) / 1e6:.2f} ms, queues {sorted(set(r['q'] for r in F))}")
by = collections.defaultdict(lambda: [0.0, 0])
for r in F:
    by[r["k"]][0] += (r["e"] - r["s"]) / 1e3; by[r["k"]][1] += 1
for k, (t, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {k[:40]:40s} n={n:5d} total={t / 1e3:8.2f} ms mean={t / n:8.1f} us")
# busy / idle / overlap
ev = sorted([(r["s"], 1) for r in F] + [(r["e"], -1) for r in F])
cur = 0; last = t0; idle = 0; one = 0; multi = 0
for t, d in ev:
    dt = t - last
    if cur == 0: idle += dt
    elif cur == 1: one += dt
    else: multi += dt
    cur += d; last = t
print(f"  idle {idle / 1e6:.2f} ms, exactly one kernel {one / 1e6:.2f} ms, two or more {multi / 1e6:.2f} ms")
# the critical chain: consecutive pivot-block launches
D = [r for r in F if r["k"].startswith("k_big_diag_reg")]
gaps = [(D[i + 1]["s"] - D[i]["e"]) / 1e3 for i in range(len(D) - 1)]
tail = gaps[-60:]
if len(D) > 1: print(f"  pivot blocks: {len(D)} launches, mean duration {sum((r['e'] - r['s']) for r in D) / len(D) / 1e3:.1f} us; gap between consecutive pivot blocks over the last 60 levels: "
      f"mean {sum(tail) / len(tail):.1f} us, min {min(tail):.1f}, max {max(tail):.1f}")
print("  schedule of the last 40 launches (start offset us, duration us, queue, kernel, grid):")
for r in F[-44:-4]:
    print(f"    {(r['s'] - t0) / 1e3:10.1f} {(r['e'] - r['s']) / 1e3:8.1f}  q{r['q']:>3s}  {r['k'][:28]:28s} {r.get('Grid_Size_X', r.get('Grid_Size', ''))}x{r.get('Grid_Size_Y', '')}")
if len(sys.argv) > 2:      # every launch of the last factorisation, one line each
    with open(sys.argv[2], "w") as f:
        for r in F:
            f.write(f"{(r['s'] - t0) / 1e3:10.1f} {(r['e'] - r['s']) / 1e3:8.1f} q{r['q']:>3s} {r['k'][:28]:28s} {r.get('Grid_Size_X', '')}x{r.get('Grid_Size_Y', '')}\n")

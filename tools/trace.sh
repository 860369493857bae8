#!/bin/bash
# rocprofv3 kernel trace of a few factor+solve passes of one workload; prints per-kernel totals and the duration histogram
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-synth_1e6}
OUT=$R/gpurun_out/trace_$WL
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python $R/tools/tune2.py $WL > $OUT/run.log 2>&1
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee $OUT/summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
ov = 0; cur_end = 0
for a, b in iv:
    if a < cur_end: ov += min(b, cur_end) - a
    cur_end = max(cur_end, b)
print(f"time with two or more kernels in flight: {ov/1e6:.2f} ms")
import os
for focus in os.environ.get("FOCUS", "k_big_assemble").split(","):
  fr = [r for r in rows if focus in r["Kernel_Name"]]
  if True:
    fr.sort(key=lambda r: int(r["Start_Timestamp"]))
    nper = len(fr) // 7 if len(fr) >= 7 else len(fr)            # 7 factorisations in tools/tune2.py
    print(f"-- {focus}: first 14 launches of the LAST factorisation (duration us, grid x/y/z in threads), then sum of the rest")
    rest = 0.0
    for i, r in enumerate(fr[-nper:]):
      d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
      if i >= 14: rest += d
      else: print(f"   #{i:3d} {d:8.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size'))} x {r.get('Grid_Size_Y', '')} x {r.get('Grid_Size_Z', '')}  wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size'))}")
    print(f"   rest: {rest/1e3:.2f} ms")
tot = sum(sum(v) for v in by.values())
print(f"total kernel time {tot/1e3:.1f} ms over {len(rows)} launches")
for k, v in sorted(by.items(), key=lambda t: -sum(t[1]))[:24]:
    v.sort()
    n = len(v)
    print(f"{k[:44]:44s} n={n:6d} total={sum(v)/1e3:9.2f} ms  mean={sum(v)/n:8.1f} us  p50={v[n//2]:8.1f} p90={v[int(n*0.9)]:8.1f} max={v[-1]:8.1f}  top5%sum={sum(v[int(n*0.95):])/1e3:8.2f} ms")
PY
rm -rf $OUT/t

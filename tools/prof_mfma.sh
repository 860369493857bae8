#!/bin/bash
# MFMA utilisation (rocprofv3 derived metric MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * SIMDs)) of the trailing
# update against the same metric of the pure-MFMA microbenchmark; separate PMC pass, kernel trace only
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-synth_1e6}
OUT=$R/gpurun_out/prof_mfma_$WL
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/micro -o pmc -- $R/tools/micro/mfma_f64_peak.bin > $OUT/micro.log 2>&1
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/bench -o pmc -- python $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-also > /dev/null 2> $OUT/bench.log
rocprofv3 --pmc LDSBankConflict --kernel-trace --output-format csv -d $OUT/lds -o pmc -- python $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-also > /dev/null 2> $OUT/lds.log
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections
def agg(d):
    tot = collections.defaultdict(list)
    for f in glob.glob(f"$OUT/{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            tot[(row["Kernel_Name"].split("(")[0], row["Counter_Name"])].append(float(row["Counter_Value"]))
    return tot
for d in ("micro", "bench", "lds"):
    for (k, c), v in sorted(agg(d).items()):
        v.sort()
        print(f"{d:6s} {k[:40]:40s} {c:18s} launches {len(v):6d}  mean {sum(v)/len(v):8.2f}  p50 {v[len(v)//2]:8.2f}  max {v[-1]:8.2f}")
# the largest trailing-update launches: utilisation next to duration and grid
rows = []
for f in glob.glob("$OUT/bench/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_big_schur" in row["Kernel_Name"] and row["Counter_Name"] == "MfmaUtil":
            rows.append((float(row["Counter_Value"]), int(row.get("End_Timestamp", 0)) - int(row.get("Start_Timestamp", 0)), row.get("Grid_Size", row.get("Grid_Size_X", "?")), row.get("Workgroup_Size", "?")))
rows.sort(key=lambda t: -t[1])
for u, d, g, w in rows[:12]: print(f"   k_big_schur  MfmaUtil {u:6.2f} %  duration {d/1e3:8.1f} us  grid {g} wg {w}")
PY
rm -rf $OUT/micro $OUT/bench $OUT/lds

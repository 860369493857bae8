#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench (and PMC passes for HBM traffic), run on the GPU box
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-lukvle1_1e4}
OUT=$R/gpurun_out/prof_$WL
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $R/bench.py --workload $WL --steps 20 --no-cpu-baseline > $OUT/bench_under_trace.json 2> $OUT/trace.log
find $OUT/trace -name "*kernel_stats*" | head -3
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -25 $f
# separate PMC passes (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2: one pass each)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o pmc -- python $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o pmc -- python $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.log
python3 - <<PY
import csv, glob, collections, json
def agg(d, name):
    files = glob.glob(f"$OUT/{d}/**/*counter_collection.csv", recursive=True)
    tot = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == name:
                k = row["Kernel_Name"].split("(")[0]
                tot[k][0] += float(row["Counter_Value"]); tot[k][1] += 1
    return tot
fe, wr = agg("pmc_fetch", "FETCH_SIZE"), agg("pmc_write", "WRITE_SIZE")
out = {}
for k in sorted(set(fe) | set(wr)):
    f, nf = fe.get(k, [0, 0]); w, nw = wr.get(k, [0, 0])
    # FETCH_SIZE/WRITE_SIZE are in KiB... rocprofv3 reports them in kilobytes; gfx950: FETCH_SIZE reads 1/2 of wide coalesced streams (MI355X_MICROARCH.md HBM)
    out[k] = dict(fetch_kb_per_launch=(f / nf if nf else None), write_kb_per_launch=(w / nw if nw else None), launches=max(nf, nw))
json.dump(out, open("$OUT/pmc_summary.json", "w"), indent=1)
for k, v in out.items(): print(k[:60], v)
PY

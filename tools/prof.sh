#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench (and PMC passes for HBM traffic), run on the GPU box
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-synth_1e6}
KERN=${2:-k_big_schur}        # kernel whose HBM traffic goes to profiles/traffic_latest.json
OUT=$R/gpurun_out/prof_$WL
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $R/bench.py --workload $WL --steps 20 --no-cpu-baseline --no-also > $OUT/bench_under_trace.json 2> $OUT/trace.log
find $OUT/trace -name "*kernel_stats*" | head -3
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -25 $f
# the dominant kernel's time as rocprofv3 sees it in the TIMED schedule (look-ahead stream on): profiles/rocprof_latest.json, from which
# bench.py computes roofline.frac_rocprof next to the hip-event figure (the two must not drift apart unseen)
python3 - <<PY
import csv, json, subprocess
rows = list(csv.DictReader(open("$OUT/kernel_stats.csv"))) if __import__("os").path.exists("$OUT/kernel_stats.csv") else []
tot = {}
for r in rows:
    tot[r["Name"].split("(")[0].replace("void ", "")] = (int(r["Calls"]), float(r["TotalDurationNs"]))
nf = max(tot.get("mi355x::k_reduce_stats", (1, 0))[0], 1)       # exactly one per factorisation
sel = {k: v for k, v in tot.items() if "$KERN" in k}
if sel:
    sh = subprocess.run(["python3", "-c", "import sys; sys.path.insert(0, '$R'); import bench; print(bench.source_hash())"], capture_output=True, text=True).stdout.strip()
    kh = subprocess.run(["python3", "-c", "import sys; sys.path.insert(0, '$R'); import bench; print(bench.kernel_code_hash('$KERN'))"], capture_output=True, text=True).stdout.strip()
    json.dump({"workload": "$WL", "kernel": "$KERN", "source_hash": sh, "kernel_code_hash": (None if kh in ("", "None") else kh), "factorisations": nf,
               "kernels": {k: {"calls_per_factorisation": c / nf, "ms_per_factorisation": ns / nf / 1e6} for k, (c, ns) in sel.items()},
               "ms_per_factorisation": sum(ns for c, ns in sel.values()) / nf / 1e6,
               "note": "rocprofv3 --kernel-trace --stats of bench.py --steps 20 (tools/prof.sh): total duration of every kernel whose name contains the dominant kernel's, per factorisation"},
              open("$OUT/rocprof_latest.json", "w"), indent=1)
PY
# separate PMC passes (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2: one pass each)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o pmc -- python $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-also > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o pmc -- python $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-also > /dev/null 2> $OUT/pmc_write.log
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
def per_factor(tot, kern):
    # totals of one kernel per FACTORISATION (k_reduce_stats runs exactly once per factorisation)
    nf = max(tot.get("mi355x::k_reduce_stats", [0, 0])[1], 1)
    key = [k for k in tot if kern in k]
    return (sum(tot[k][0] for k in key) / nf, sum(tot[k][1] for k in key) / nf) if key else (None, None)
fk, fl = per_factor(fe, "$KERN"); wk, wl_ = per_factor(wr, "$KERN")
if fk is not None and wk is not None:
    import subprocess
    sh = subprocess.run(["python3", "-c", "import sys; sys.path.insert(0, '$R'); import bench; print(bench.source_hash())"], capture_output=True, text=True).stdout.strip()
    kh = subprocess.run(["python3", "-c", "import sys; sys.path.insert(0, '$R'); import bench; print(bench.kernel_code_hash('$KERN'))"], capture_output=True, text=True).stdout.strip()
    json.dump({"workload": "$WL", "kernel": "$KERN", "source_hash": sh, "kernel_code_hash": (None if kh in ("", "None") else kh), "launches_per_factorisation": fl,
               "fetch_bytes_per_factorisation_raw": fk * 1024, "write_bytes_per_factorisation": wk * 1024,
               "hbm_bytes_per_factorisation": 2 * fk * 1024 + wk * 1024,
               "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/prof.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md "
                       "(gfx950 reports 1/2 of coalesced streaming reads; 8-byte-per-lane patterns are uncalibrated => upper estimate); WRITE_SIZE as "
                       "reported; Infinity-Cache hits are counted.  bench.py divides by its own launches_per_factor_solve."},
              open("$OUT/traffic_latest.json", "w"), indent=1)
out = {}
for k in sorted(set(fe) | set(wr)):
    f, nf = fe.get(k, [0, 0]); w, nw = wr.get(k, [0, 0])
    # FETCH_SIZE/WRITE_SIZE are in KiB... rocprofv3 reports them in kilobytes; gfx950: FETCH_SIZE reads 1/2 of wide coalesced streams (MI355X_MICROARCH.md HBM)
    out[k] = dict(fetch_kb_per_launch=(f / nf if nf else None), write_kb_per_launch=(w / nw if nw else None), launches=max(nf, nw))
json.dump(out, open("$OUT/pmc_summary.json", "w"), indent=1)
for k, v in out.items(): print(k[:60], v)
PY

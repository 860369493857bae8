#!/bin/bash
# The round's closing measurements in ONE gpu call (the budget left for it: ~5 GPU-minutes): GPU suite, rocprofv3 kernel stats + PMC traffic of the default
# workload, quick bench lines (no CPU baseline: the driver's own run carries it), launch timelines.   usage (on the GPU box, repo root): tools/final_pass.sh r05c
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; grep -n "passed\|failed" $O/gpu_tests.txt | tail -2
tools/prof.sh synth_1e6 k_big_schur > $O/prof_synth_1e6.log 2>&1
for f in kernel_stats.csv pmc_summary.json traffic_latest.json rocprof_latest.json bench_under_trace.json; do cp gpurun_out/prof_synth_1e6/$f $O/synth_1e6_$f 2>/dev/null; done
cp gpurun_out/prof_synth_1e6/traffic_latest.json gpurun_out/prof_synth_1e6/rocprof_latest.json profiles/ 2>/dev/null
python bench.py --no-cpu-baseline --no-e2e > $O/bench_synth_1e6_nocpu.json 2> $O/bench_synth_1e6_nocpu.err; cut -c1-400 $O/bench_synth_1e6_nocpu.json
for wl in grid_1e5 lukvle1_1e6 lukvle1_1e4 mbndry1_100; do python bench.py --workload $wl --no-cpu-baseline --no-e2e --no-also > $O/bench_${wl}_nocpu.json 2> $O/bench_${wl}_nocpu.err; cut -c1-200 $O/bench_${wl}_nocpu.json; done
python bench.py --e2e-only LukVlE1:1000000:1 > $O/e2e_lukvle1_1e6.json 2> $O/e2e_lukvle1_1e6.err; cut -c1-600 $O/e2e_lukvle1_1e6.json
tools/timeline.sh synth_1e6 $TAG > $O/synth_1e6_timeline.txt 2>&1; cp gpurun_out/timeline_synth_1e6_$TAG/all_launches.txt $O/synth_1e6_all_launches.txt 2>/dev/null
[ -f $R/.dev_pivstat/mb3d_50.npz ] && { tools/timeline.sh npz:$R/.dev_pivstat/mb3d_50.npz $TAG > $O/mb3d_50_timeline.txt 2>&1; cp gpurun_out/timeline_mb3d_50_$TAG/all_launches.txt $O/mb3d_50_all_launches.txt 2>/dev/null; }
ls $O | head -40

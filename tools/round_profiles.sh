#!/bin/bash
# Everything the round's DESIGN.md / profiles/ numbers come from, in one GPU call:
#   bench lines of the four workloads, rocprofv3 kernel stats + PMC traffic of the default workload, launch timeline, pivot-block phases,
#   host analysis times, the many-handles stress.
# usage (on the GPU box, from the repo root):  tools/round_profiles.sh r02
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
# PMC traffic first: bench.py only reports roofline.traffic from a profiles/traffic_latest.json measured with the CURRENT kernel sources
tools/prof.sh synth_1e6 k_big_schur > $O/prof_synth_1e6.log 2>&1
cp gpurun_out/prof_synth_1e6/kernel_stats.csv $O/synth_1e6_kernel_stats.csv 2>/dev/null
cp gpurun_out/prof_synth_1e6/pmc_summary.json $O/synth_1e6_pmc_summary.json 2>/dev/null
cp gpurun_out/prof_synth_1e6/traffic_latest.json $O/traffic_latest.json 2>/dev/null
cp gpurun_out/prof_synth_1e6/traffic_latest.json profiles/traffic_latest.json 2>/dev/null      # (this copy of the tree; commit the one under $O)
cp gpurun_out/prof_synth_1e6/rocprof_latest.json $O/rocprof_latest.json 2>/dev/null
cp gpurun_out/prof_synth_1e6/rocprof_latest.json profiles/rocprof_latest.json 2>/dev/null
cp gpurun_out/prof_synth_1e6/bench_under_trace.json $O/synth_1e6_bench_under_rocprof.json 2>/dev/null
python bench.py > $O/bench_synth_1e6.json 2> $O/bench_synth_1e6.err
for wl in grid_1e5 lukvle1_1e6 lukvle1_1e4 mbndry1_100; do python bench.py --workload $wl --no-e2e > $O/bench_$wl.json 2> $O/bench_$wl.err; done
python bench.py --e2e-only MBndryCntrl_3D:30:1,16 > $O/e2e_mbndry3d_30.json 2> $O/e2e_mbndry3d_30.err
(make -s -C tools/micro >/dev/null 2>&1; tools/prof_mfma.sh synth_1e6 > $O/synth_1e6_mfma_util.txt 2>&1)
python -m pytest tests/test_multigpu_gpu.py -q > $O/multigpu_gpu_tests.log 2>&1
tools/timeline.sh synth_1e6 $TAG > $O/synth_1e6_timeline.txt 2>&1
python tools/clocks.py synth_1e6 > $O/synth_1e6_pivot_block_phases.txt 2>&1
tools/prof.sh lukvle1_1e6 k_front_reg > $O/prof_lukvle1_1e6.log 2>&1
cp gpurun_out/prof_lukvle1_1e6/kernel_stats.csv $O/lukvle1_1e6_kernel_stats.csv 2>/dev/null
cp gpurun_out/prof_lukvle1_1e6/pmc_summary.json $O/lukvle1_1e6_pmc_summary.json 2>/dev/null
python tools/analysis_time.py lukvle1_1e6 synth_1e6 > $O/analysis_time.txt 2>&1
(python tools/match_time.py lukvle1_1e6 synth_1e6 2>&1 | grep -v "^\[mi355x_kkt\]   \|look-ahead\|data-flow\|grouped\|chain look") > $O/matching_scaling_device.txt 2>&1
timeout 120 python tools/stress_handles.py 60 > $O/stress_handles.txt 2>&1
ls -la $O

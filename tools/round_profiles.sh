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
# round 6: several right-hand sides through the solve contexts, the cost of a delayed-pivot edit, the contribution-block plan on the 3-D family (if the recorded
# systems are there: .dev_pivstat/ is scratch), two ranks on the shared device (bench line with its e2e leg), the multi-rank Ipopt runs
python tools/multirhs_time.py synth_1e6 grid_1e5 lukvle1_1e6 mbndry1_100 > $O/multirhs_time.json 2> $O/multirhs_time.err
python tools/delay_cost.py synth_1e6 100 > $O/delay_cost_synth_1e6.json 2> /dev/null
for N in 50 78 100; do [ -f $R/.dev_pivstat/mb3d_$N.npz ] && for m in 0 1; do MI355X_KKT_RECYCLE=$m python tools/mem_report.py npz:$R/.dev_pivstat/mb3d_$N.npz 2>/dev/null | tail -1 >> $O/mem_report_recycle$m.txt; done; done
MI355X_KKT_BENCH_SHARED=1 timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 2>/dev/null | grep '^{"metric' > $O/bench_2ranks_shared_device.json
python -m pytest tests/test_e2e_multirank.py -q > $O/e2e_multirank_tests.log 2>&1
ls -la $O

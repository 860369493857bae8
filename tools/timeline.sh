#!/bin/bash
# rocprofv3 kernel timeline of the LAST factorisation of tools/factor_loop.py: per-kernel totals, overlap between queues, and the
# launch-by-launch schedule of the top of the tree (the separator chains).  usage: tools/timeline.sh [workload] [tag]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-synth_1e6}; TAG=${2:-run}
WLN=$(basename ${WL%.npz}); OUT=$R/gpurun_out/timeline_${WLN#npz:}_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python $R/tools/factor_loop.py $WL 4 > $OUT/run.log 2>&1
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python3 $R/tools/timeline_report.py "$f" $OUT/all_launches.txt | tee $OUT/summary.txt
python3 $R/tools/solve_report.py "$f" $OUT/solve_launches.txt > $OUT/solve_summary.txt
rm -rf $OUT/t

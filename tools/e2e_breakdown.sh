#!/bin/bash
# Ipopt's own timing statistics (print_timing_statistics) of the north-star instance with the three MI355X routes
cd /tmp
D=/root/repo/oracle/_ref
printf "print_timing_statistics yes\n" > /tmp/timing.opt
export MKL_NUM_THREADS=1 OMP_NUM_THREADS=1
P=${1:-LukVlE1}; N=${2:-1000000}
for s in mi355x mi355x-aug ${3:-}; do
  $D/ipopt_mi355x_driver $P $N --solver $s --quiet > /dev/null 2>&1
  echo "== $s"
  $D/ipopt_mi355x_driver $P $N --solver $s --optfile /tmp/timing.opt 2>&1 | grep -E "DRIVER_SUMMARY|AUG_STATS|PDSystemSolver|LinearSystem|StdAugSystem|TaskTotal|OverallAlgorithm|UpdateHessian|ComputeSearchDirection|Number of Iterations|FunctionEvaluations|InitializeIterates|CheckConvergence|OutputIteration|ComputeAcceptableTrialPoint" | cut -c1-240
done

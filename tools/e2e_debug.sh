#!/bin/bash
# usage: tools/e2e_debug.sh PROBLEM N [extra driver args]   -- one end-to-end run of the reference host with the MI355X backend, backend messages on
cd /tmp
export MKL_NUM_THREADS=1 OMP_NUM_THREADS=1
P=$1; N=$2; shift 2
$GRAFT_REPO_ROOT/oracle/_ref/ipopt_mi355x_driver $P $N --solver mi355x "$@" 2>&1

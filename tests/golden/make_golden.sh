#!/bin/bash
# Regenerates tests/golden/ from the REFERENCE ITSELF (unmodified coin-or/Ipopt 3.14.15 built by
# oracle/ref_build.mk, linear solver = the reference's PardisoMKLSolverInterface on oneMKL -- the only
# CPU backend available offline).  Needs /root/reference; run in the build container:
#     make -f oracle/ref_build.mk -j8 && bash tests/golden/make_golden.sh
# Products:
#   *.kktrec   every call that crossed the SparseSymLinearSolverInterface boundary during the run
#              (structure, values, rhs, returned status / inertia / solution); reader: oracle/kkt_oracle.py
#   *.iters    the iteration table of the reference run (iter objective inf_pr inf_du lg(mu) lg(rg) ls)
#   *.summary  DRIVER_SUMMARY json (iteration count, objective, reference timers on the build container)
set -e
cd "$(dirname "$0")"
D=../../oracle/_ref
export MKL_NUM_THREADS=1 OMP_NUM_THREADS=1
# tail <name> <problem> <N> <K>: the LAST K boundary calls of the run (late-barrier systems: Sigma spanning many decades, delta_c active --
# IpPDPerturbationHandler.cpp:467-470) into <name>_late.kktrec; a first run counts the calls (RECORD_CALLS), the second skips all but the last K
tail_rec() {
  name=$1; p=$2; n=$3; k=$4
  calls=$($D/ref_driver $p $n --record /dev/null --max-records 0 --quiet | grep RECORD_CALLS | awk '{print $2}')
  $D/ref_driver $p $n --record ${name}_late.kktrec --skip-records $((calls - k)) --quiet > /dev/null
  echo "${name}_late: calls $((calls - k + 1))..$calls of $calls"
}
run() {  # name problem N record?
  name=$1; p=$2; n=$3; rec=$4
  if [ "$rec" = rec ]; then $D/ref_driver $p $n --record $name.kktrec > /tmp/$name.log;
  elif [ "${rec#rec}" != "$rec" ]; then $D/ref_driver $p $n --record $name.kktrec --max-records ${rec#rec} > /tmp/$name.log;      # recN: the first N boundary calls only
  else $D/ref_driver $p $n > /tmp/$name.log; fi
  grep -E "^ +[0-9]+r? " /tmp/$name.log | awk '{print $1, $2, $3, $4, $5, $7, $10}' > $name.iters
  grep DRIVER_SUMMARY /tmp/$name.log | sed 's/^DRIVER_SUMMARY //' > $name.summary
  echo "$name: $(wc -l < $name.iters) iteration lines"
}
run hs071 hs071 0 rec
run lukvle1_100 LukVlE1 100 rec
run mbndry1_8 MBndryCntrl1 8 rec
# BASELINE.json configs[1] and [2]: iteration tables + the first four calls across the boundary (a full recording is 20 MB)
run lukvle1_10000 LukVlE1 10000 rec4
run mbndry1_100 MBndryCntrl1 100 rec4
# ... and the LAST four calls of configs[1], configs[2] and of MBndryCntrl2 N = 100 (the problem whose factorisations delay pivots)
tail_rec lukvle1_10000 LukVlE1 10000 4
tail_rec mbndry1_100 MBndryCntrl1 100 4
tail_rec mbndry2_100 MBndryCntrl2 100 4
run lukvle1_1000000 LukVlE1 1000000 norec
# more problem classes of examples/ScalableProblems (inequalities, other PDE controls, 3-D): iteration tables only
run lukvli1_10000 LukVlI1 10000 norec
run lukvle5_10000 LukVlE5 10000 norec
run mbndry2_100 MBndryCntrl2 100 norec
run mdist1_100 MDistCntrl1 100 norec
run mbndry3d_12 MBndryCntrl_3D 12 norec
# the first boundary call of a small 3-D instance: its separator chains have the dangling side children that finish_analysis 9b hangs down the chain
# (tests/test_symbolic.py::test_side_children_of_chain_links_hang_down_the_chain)
run mbndry3d_14 MBndryCntrl_3D 14 rec1
run mbndry3d_30 MBndryCntrl_3D 30 norec      # 3-D separators (fronts of ~2 000 rows at KKT dimension 50 600): SURVEY 8(d)-5's MFMA-bound family at a size the CPU run takes 17 s for
run mbndry1_300 MBndryCntrl1 300 norec
# the CUTEst-style ~10^6 stand-in of BASELINE.json configs[4] (n = 492 800, m = 490 000; examples/ScalableProblems/solve_problem.cpp:28-91),
# 8 MKL threads (the iteration table does not depend on the thread count; one thread takes a minute per run)
MKL_NUM_THREADS=8 OMP_NUM_THREADS=8 run mbndry1_700 MBndryCntrl1 700 norec
ls -la
# the PDSystemSolver boundary: pieces of the 8-block system, right-hand side and result of the reference's
# PDFullSpaceSolver::Solve, every call of hs071 and the first 8 of LukVlI1 n = 20 (bounds on every variable); reader: oracle/pd_oracle.py
$D/ref_driver hs071 0 --record-pd hs071.pdrec --quiet > /dev/null
$D/ref_driver LukVlI1 20 --record-pd lukvli1_20.pdrec --max-records 8 --quiet > /dev/null
# SURVEY 8(d)-5's MFMA-bound 3-D family at a larger size (n = 125 000, m = 110 592, KKT dimension 235 592; fronts up to ~5 600 rows): one minute on 8 MKL threads
MKL_NUM_THREADS=8 OMP_NUM_THREADS=8 run mbndry3d_50 MBndryCntrl_3D 50 norec
# SURVEY 8(d)-5's 3-D instance itself: MBndryCntrl_3D N = 78 (n = 80^3 = 512 000, m = 78^3 = 474 552, KKT dimension 986 552; examples/ScalableProblems/solve_problem.cpp:56):
# 16 iterations, 13 minutes on 8 MKL threads (PDSystemSolverTotal 727 s of the 775 s)
MKL_NUM_THREADS=8 OMP_NUM_THREADS=8 run mbndry3d_78 MBndryCntrl_3D 78 norec
# round 6: the same 3-D family at N = 100 (n = 102^3 = 1 061 208, m = 10^6, KKT dimension 2 060 000; fronts of up to 22 448 rows, 19.2 TFlop per factorisation with our
# ordering): 16 iterations, 1 h 35 min on 6 MKL threads (PDSystemSolverTotal 5 281 s of 5 679 s) -- the instance the contribution-block recycling of round 6 makes room for
MKL_NUM_THREADS=6 OMP_NUM_THREADS=6 run mbndry3d_100 MBndryCntrl_3D 100 norec
# round 6: EVERY problem class the reference's own driver registers (examples/ScalableProblems/solve_problem.cpp:28-91) that its CPU run solves: iteration tables
# only, 4 MKL threads, the whole sweep in four minutes.  (MPara5_1 and MPara5_2_1 at N = 40 end in "Maximum Number of Iterations Exceeded" and MPara5_2_2 in
# "Error in step computation" with the reference's own MKL PARDISO path: no golden for those.  LukVl3 / LukVl4 want N + 2 divisible by 4.)
export MKL_NUM_THREADS=4 OMP_NUM_THREADS=4
for k in 2 6 7; do run lukvle${k}_10000 LukVlE$k 10000 norec; run lukvli${k}_10000 LukVlI$k 10000 norec; done
for k in 3 4; do run lukvle${k}_9998 LukVlE$k 9998 norec; run lukvli${k}_9998 LukVlI$k 9998 norec; done
run lukvli5_10000 LukVlI5 10000 norec
for k in 3 4 5 6 7 8; do run mbndry${k}_100 MBndryCntrl$k 100 norec; done
for k in 2 3 3a 4 5 6a 4a 5a 6; do run mdist${k}_100 MDistCntrl$k 100 norec; done
run mbndry3d27_12 MBndryCntrl_3D_27 12 norec; run mbndry3d27bt_12 MBndryCntrl_3D_27BT 12 norec; run mbndry3dsin_12 MBndryCntrl_3Dsin 12 norec
run mpara5_2_3_40 MPara5_2_3 40 norec

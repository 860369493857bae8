#!/bin/bash
# End-to-end Ipopt timing on the GPU box: reference host + MI355X backend vs reference host + MKL PARDISO (1 and 16 threads)
cd /tmp
D=/root/repo/oracle/_ref
for spec in "$@"; do
  p=${spec%%:*}; n=${spec##*:}
  export MKL_NUM_THREADS=1 OMP_NUM_THREADS=1   # host BLAS-1 of Ipopt itself: 1 thread (256-thread MKL on tiny vectors is pathological)
  $D/ipopt_mi355x_driver $p $n --solver mi355x --quiet > /dev/null 2>&1   # warm-up (page-in, clocks)
  echo "== $p $n"
  $D/ipopt_mi355x_driver $p $n --solver mi355x --quiet 2>&1 | grep DRIVER_SUMMARY | sed 's/DRIVER_SUMMARY/mi355x      /'
  MKL_NUM_THREADS=1 OMP_NUM_THREADS=1 $D/ipopt_mi355x_driver $p $n --solver pardisomkl --quiet 2>&1 | grep DRIVER_SUMMARY | sed 's/DRIVER_SUMMARY/mkl_1thread /'
  MKL_NUM_THREADS=16 OMP_NUM_THREADS=16 $D/ipopt_mi355x_driver $p $n --solver pardisomkl --quiet 2>&1 | grep DRIVER_SUMMARY | sed 's/DRIVER_SUMMARY/mkl_16thread/'
done

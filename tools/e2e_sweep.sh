#!/bin/bash
cd /tmp
D=/root/repo/oracle/_ref
export MKL_NUM_THREADS=1 OMP_NUM_THREADS=1
for spec in "$@"; do
  p=${spec%%:*}; n=${spec##*:}
  a=$(timeout 600 $D/ipopt_mi355x_driver $p $n --solver mi355x --quiet 2>&1 | grep DRIVER_SUMMARY | sed 's/DRIVER_SUMMARY //')
  b=$(MKL_NUM_THREADS=8 timeout 900 $D/ipopt_mi355x_driver $p $n --solver pardisomkl --quiet 2>&1 | grep DRIVER_SUMMARY | sed 's/DRIVER_SUMMARY //')
  python3 - "$p" "$n" "$a" "$b" <<'PY'
import sys, json
p, n, a, b = sys.argv[1:5]
try: A = json.loads(a)
except Exception: A = None
try: B = json.loads(b)
except Exception: B = None
f = lambda J: "FAILED" if J is None else f"status {J['status']} iters {J['iterations']} obj {J['objective']:.10e} PDTotal {J['PDSystemSolverTotal']:.3f}s overall {J['OverallAlgorithm']:.3f}s"
print(f"{p:16s} {n:>8s} | mi355x: {f(A)} | mkl(8thr): {f(B)}")
PY
done

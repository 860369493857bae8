#!/bin/bash
# end-to-end Ipopt runs (reference host + MI355X backend vs reference host + MKL PARDISO) on the GPU box
cd /tmp
D=/root/repo/oracle/_ref
for spec in "$@"; do
  set -- $spec
  p=${spec%%:*}; n=${spec##*:}
  for s in mi355x pardisomkl; do
    timeout 600 $D/ipopt_mi355x_driver $p $n --solver $s > /tmp/e2e_${p}_${n}_${s}.log 2>&1
    echo "== $p $n $s rc=$?"
    grep -E "^ +[0-9]+r? " /tmp/e2e_${p}_${n}_${s}.log | awk '{print $1, $2, $3, $4, $5, $7, $10}' | tail -40 > /tmp/e2e_${p}_${n}_${s}.iters
    grep -E "EXIT|DRIVER_SUMMARY" /tmp/e2e_${p}_${n}_${s}.log
  done
  if diff -q /tmp/e2e_${p}_${n}_mi355x.iters /tmp/e2e_${p}_${n}_pardisomkl.iters >/dev/null; then echo "ITERATION LOGS IDENTICAL ($p $n)"; else echo "ITERATION LOGS DIFFER ($p $n)"; diff /tmp/e2e_${p}_${n}_mi355x.iters /tmp/e2e_${p}_${n}_pardisomkl.iters | head -10; fi
done

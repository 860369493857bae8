#!/bin/bash
# A/B of environment switches inside ONE gpu call (boxes differ by several per cent): usage tools/ab.sh OUTDIR WORKLOAD "ENV1" "ENV2" ...
O=$1; WL=$2; shift 2
mkdir -p $O
i=0
for e in "$@"; do
  env $e timeout 300 python tools/factor_loop.py $WL 6 > $O/ab_${WL}_$i.log 2>&1
  echo "[$e] $(grep 'rep 5' $O/ab_${WL}_$i.log) | $(tail -1 $O/ab_${WL}_$i.log | cut -c1-120)"
  i=$((i+1))
done

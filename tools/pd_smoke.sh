cd /tmp
D=/root/repo/oracle/_ref
export MKL_NUM_THREADS=1 OMP_NUM_THREADS=1
for spec in hs071:0 LukVlE1:100 LukVlE1:10000 MBndryCntrl1:8 MBndryCntrl1:100 LukVlI1:10000; do
  p=${spec%%:*}; n=${spec##*:}
  for s in mi355x mi355x-pd; do
    timeout 300 $D/ipopt_mi355x_driver $p $n --solver $s --quiet 2>&1 | grep -E "DRIVER_SUMMARY|rror|xception" | cut -c1-330
  done
done

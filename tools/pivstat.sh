#!/bin/bash
# Development aid: builds an instrumented copy of the library (-DMI355X_PIVSTAT: cycles of the fast / slow iterations of the
# pivot loop in the chain's pivot blocks; -DMI355X_PIVSTAT_COUNT adds per-decision counters, which perturb the timing) into
# .dev_pivstat/ and prints the statistics of three factorisations.  Run the second half on the GPU box:
#   bash tools/pivstat.sh build [count]  &&  gpurun -- 'bash tools/pivstat.sh run synth_1e6'
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p .dev_pivstat
  DEFS="-DMI355X_PIVSTAT"; [ "$2" = count ] && DEFS="$DEFS -DMI355X_PIVSTAT_COUNT"
  make -s -C ipopt_amd
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $DEFS --offload-arch=gfx950 -c ipopt_amd/csrc/numeric.hip -o .dev_pivstat/numeric.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o .dev_pivstat/libmi355x_kkt.so ipopt_amd/lib/symbolic.o ipopt_amd/lib/matching_scaling.o ipopt_amd/lib/api.o .dev_pivstat/numeric.o ipopt_amd/lib/ma97_abi.o
else
  cp .dev_pivstat/libmi355x_kkt.so ipopt_amd/lib/libmi355x_kkt.so      # (the box's copy of the tree is scratch)
  python tools/clocks.py "${2:-synth_1e6}" 2>&1 | grep -E "PIVSTAT|cycles"
fi

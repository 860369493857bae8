#!/bin/bash
# the four single-GPU bench lines (with the CPU baseline) -> gpurun_out/bench_<workload>.json
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
for wl in lukvle1_1e4 lukvle1_1e6 grid_1e5; do
  python $R/bench.py --workload $wl > $R/gpurun_out/bench_$wl.json 2> $R/gpurun_out/bench_$wl.err
done
python $R/bench.py > $R/gpurun_out/bench_synth_1e6.json 2> $R/gpurun_out/bench_synth_1e6.err
for wl in lukvle1_1e4 lukvle1_1e6 grid_1e5 synth_1e6; do
  python3 - $R/gpurun_out/bench_$wl.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = j["roofline"]; c = j.get("cpu_baseline", {})
print(f'{j["config"]["workload"]:12s} {j["ms_per_step"]:8.3f} ms/step  {j["value"]:8.1f} GFLOP/s  factor {j["device_ms"]["factor"]:7.3f} solve {j["device_ms"]["solve"]:6.3f}  dom {r["kernel"]} {r["achieved"]:.1f} {r["unit"]} frac {r["frac"]:.4f}  cpu {c.get("ms_per_step")} ms ({c.get("cores")} thr)  analyse {j["analyse_s"]:.2f}s')
PY
done

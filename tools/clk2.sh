#!/bin/bash
cat > /tmp/loop.py <<'PY'
import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ipopt_amd, bench
n, r, c, v, neg = bench.make_workload("lukvle1_1e4")
s = ipopt_amd.KKTSolver(); s.initialize_structure(n, r, c, vals=v); s.values()[:] = v
t0 = time.time(); k = 0
while time.time() - t0 < 6:
    s.multi_solve(True, None); k += 1
print("factorizations/s", k / 6)
PY
python /tmp/loop.py &
sleep 3
rocm-smi --showclocks 2>&1 | grep -E "sclk|mclk"
rocm-smi -a 2>&1 | grep -iE "sclk|performance level|power" | head -12
wait
cat > /tmp/fma.hip <<'CPP'
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain(float* out, int n) { float x = out[0]; for (int i = 0; i < n; ++i) x = fmaf(x, 1.0000001f, 1e-9f); out[0] = x; }
__global__ void chain_d(double* out, int n) { double x = out[0]; for (int i = 0; i < n; ++i) x = fma(x, 1.0000001, 1e-9); out[0] = x; }
int main() { float* d; double* dd; hipMalloc(&d, 4); hipMalloc(&dd, 8); hipMemset(d, 0, 4); hipMemset(dd, 0, 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) { int n = 2000000; hipEventRecord(a); hipLaunchKernelGGL(chain, 1, 64, 0, 0, d, n); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
    printf("f32 chain: %d dependent fma in %.3f ms -> %.2f ns per fma\n", n, ms, ms * 1e6 / n); }
  for (int rep = 0; rep < 2; ++rep) { int n = 2000000; hipEventRecord(a); hipLaunchKernelGGL(chain_d, 1, 64, 0, 0, dd, n); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
    printf("f64 chain: %d dependent fma in %.3f ms -> %.2f ns per fma\n", n, ms, ms * 1e6 / n); }
  for (int rep = 0; rep < 2; ++rep) { int n = 2000000; hipEventRecord(a); hipLaunchKernelGGL(chain, 1024, 256, 0, 0, d, n); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
    printf("f32 chain, 1024x256 threads: %.3f ms -> %.2f ns per fma\n", ms, ms * 1e6 / n); }
  return 0; }
CPP
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 fma.hip -o fma && ./fma

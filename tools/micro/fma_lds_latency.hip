// Micro-benchmark (development aid): what a lone wavefront pays per step of a 4 x 4 register-tile fp64 product fed from LDS.
//   variant 0: 16 v_fmac_f64 per step, operands in registers (no LDS)           -> cycles per FMA
//   variant 1: operands loaded from LDS at the top of every step (no prefetch)   -> exposed LDS latency
//   variant 2: operands of step q+1 requested before the FMAs of step q          -> software pipelined
//   variant 3: one v_mfma_f64_16x16x4_f64 per step, operands from LDS, dependent accumulator
//   variant 4: same, operands prefetched one step ahead
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/fma_lds_latency.hip -o tools/micro/fma_lds_latency.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4 __attribute__((ext_vector_type(4)));
template <int VAR> __global__ void kern(double* out, unsigned long long* cyc, int steps)
{
    __shared__ double A[64 * 65], B[64 * 65];
    const int tid = threadIdx.x;
    for (int i = tid; i < 64 * 65; i += blockDim.x) { A[i] = 1e-3 * (i % 17); B[i] = 1e-3 * (i % 13); }
    __syncthreads();
    const int r0 = 4 * (tid & 15), c0 = 4 * ((tid >> 4) & 15);
    double acc[4][4];
    for (int x = 0; x < 4; ++x) for (int y = 0; y < 4; ++y) acc[x][y] = 0.0;
    v4 macc = (v4){0, 0, 0, 0};
    const unsigned long long t0 = clock64();
    if (VAR == 0) {
        double lv[4] = {A[r0], A[r0 + 1], A[r0 + 2], A[r0 + 3]}, wv[4] = {B[c0], B[c0 + 1], B[c0 + 2], B[c0 + 3]};
        for (int q = 0; q < steps; ++q) {
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = fma(lv[x], wv[y], acc[x][y]);
            asm volatile("" : "+v"(lv[0]), "+v"(wv[0]));
        }
    } else if (VAR == 1) {
        for (int q = 0; q < steps; ++q) {
            double lv[4], wv[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) { lv[x] = A[r0 + x + (q & 63) * 65]; wv[x] = B[c0 + x + (q & 63) * 65]; }
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = fma(lv[x], wv[y], acc[x][y]);
        }
    } else if (VAR == 2) {
        double lv[4], wv[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) { lv[x] = A[r0 + x]; wv[x] = B[c0 + x]; }
        for (int q = 0; q < steps; ++q) {
            double ln[4], wn[4];
            const int qn = (q + 1) & 63;
#pragma unroll
            for (int x = 0; x < 4; ++x) { ln[x] = A[r0 + x + qn * 65]; wn[x] = B[c0 + x + qn * 65]; }
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = fma(lv[x], wv[y], acc[x][y]);
#pragma unroll
            for (int x = 0; x < 4; ++x) { lv[x] = ln[x]; wv[x] = wn[x]; }
        }
    } else if (VAR == 3) {
        const int l15 = tid & 15, l4 = (tid >> 4) & 3;
        for (int q = 0; q < steps; ++q) macc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[l15 + (4 * (q & 15) + l4) * 65], B[l15 + (4 * (q & 15) + l4) * 65], macc, 0, 0, 0);
    } else {
        const int l15 = tid & 15, l4 = (tid >> 4) & 3;
        double a = A[l15 + l4 * 65], b = B[l15 + l4 * 65];
        for (int q = 0; q < steps; ++q) {
            const double an = A[l15 + (4 * ((q + 1) & 15) + l4) * 65], bn = B[l15 + (4 * ((q + 1) & 15) + l4) * 65];
            macc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, macc, 0, 0, 0);
            a = an; b = bn;
        }
    }
    const unsigned long long t1 = clock64();
    double s = macc[0] + macc[1] + macc[2] + macc[3];
    for (int x = 0; x < 4; ++x) for (int y = 0; y < 4; ++y) s += acc[x][y];
    out[blockIdx.x * blockDim.x + tid] = s;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int VAR> void run(const char* what, int threads, int steps)
{
    double* out; unsigned long long* cyc; unsigned long long h = 0;
    hipMalloc(&out, 8 * 1024); hipMalloc(&cyc, 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern<VAR>, dim3(1), dim3(threads), 0, 0, out, cyc, steps);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-64s threads %4d: %7.1f cycles per step\n", what, threads, (double)h / steps);
    hipFree(out); hipFree(cyc);
}
int main()
{
    for (int th : {64, 256}) {
        run<0>("16 v_fmac_f64, register operands", th, 4096);
        run<1>("16 v_fmac_f64, 8 LDS operands loaded at the top of the step", th, 4096);
        run<2>("16 v_fmac_f64, LDS operands prefetched one step ahead", th, 4096);
        run<3>("1 v_mfma_f64_16x16x4, LDS operands, dependent accumulator", th, 4096);
        run<4>("1 v_mfma_f64_16x16x4, LDS operands prefetched one step ahead", th, 4096);
    }
    return 0;
}

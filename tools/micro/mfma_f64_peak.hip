// microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate on gfx950 (the roofline `peak` of the frontal update)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a, double b)
{
    v4f64 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4f64){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> void run(int wgs_per_cu, const char* name)
{
    int ncu = 256; double* out; hipMalloc(&out, sizeof(double) * 256 * ncu * 16);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<ncu * wgs_per_cu, 256>>>(out, 100, 1.0, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0); k<NACC><<<ncu * wgs_per_cu, 256>>>(out, iters, 1.0, 1.0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)ncu * wgs_per_cu * 4 * iters * NACC * 2048.0;
    printf("%s nacc=%d wg/cu=%d: %.2f ms  %.1f TFLOP/s\n", name, NACC, wgs_per_cu, ms, flops / ms / 1e9);
    hipFree(out);
}
int main() { run<4>(1, "mfma_f64_16x16x4"); run<8>(1, "mfma_f64_16x16x4"); run<4>(2, "mfma_f64_16x16x4"); run<4>(4, "mfma_f64_16x16x4"); run<16>(1, "mfma_f64_16x16x4"); return 0; }

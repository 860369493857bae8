// microbenchmark: the latencies a serial pivot chain is made of, on ONE workgroup of 256 threads (4 waves, one per SIMD):
//   (a) publish -> s_waitcnt -> s_barrier -> ds_read -> s_waitcnt round trip
//   (b) dependent v_fma_f64 chain, (c) v_rcp_f64 + two Newton steps, (d) v_cmp -> ballot -> scalar branch,
//   (e) 16 independent v_fma_f64, (f) wave max by DPP + readlane
// cycles per iteration by s_memtime over ITERS iterations (thread 0 of wave 0).
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 2000
__device__ __forceinline__ double dpp_f64(double x, int) { return x; }
template <int CTRL> __device__ __forceinline__ double dppf(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rl(double x, int l) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l)); }
__device__ __forceinline__ double wave_max_all(double x)
{
    x = fmax(x, dppf<0xB1>(x)); x = fmax(x, dppf<0x4E>(x)); x = fmax(x, dppf<0x141>(x)); x = fmax(x, dppf<0x140>(x));
    return fmax(fmax(rl(x, 0), rl(x, 16)), fmax(rl(x, 32), rl(x, 48)));
}

__global__ __launch_bounds__(256) void k(double* out, long long* cyc, double seed)
{
    __shared__ double buf[2][256];
    const int tid = threadIdx.x;
    double v = seed + tid * 1e-3;
    long long t0, t1;
    // (a) LDS publish/barrier/read round trip
    __syncthreads();
    t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
        if ((tid >> 4) == (it & 15)) buf[it & 1][tid & 15] = v;
        __syncthreads();
        v += buf[it & 1][tid & 15];
    }
    t1 = clock64(); if (tid == 0) cyc[0] = t1 - t0;
    // (a2) same without the barrier (single-wave semantics: ds_write -> ds_read dependency only)
    t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
        buf[it & 1][tid] = v;
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
        v += buf[it & 1][tid ^ 1];
    }
    t1 = clock64(); if (tid == 0) cyc[1] = t1 - t0;
    // (a3) barrier only
    t0 = clock64();
    for (int it = 0; it < ITERS; ++it) { __syncthreads(); asm volatile("" ::: "memory"); }
    t1 = clock64(); if (tid == 0) cyc[2] = t1 - t0;
    // (b) dependent fma chain
    double a = v * 1e-9, b = 1.0000001;
    t0 = clock64();
    for (int it = 0; it < ITERS; ++it) { a = fma(a, b, 1e-9); asm volatile("" : "+v"(a)); }
    t1 = clock64(); if (tid == 0) cyc[3] = t1 - t0;
    // (c) rcp + 2 Newton, dependent on the previous result
    double d = 1.0 + a * 1e-12;
    t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
        double r = __builtin_amdgcn_rcp(d); double e = fma(-d, r, 1.0); r = fma(r, e, r); e = fma(-d, r, 1.0); r = fma(r, e, r);
        d = r + 1e-3; asm volatile("" : "+v"(d));
    }
    t1 = clock64(); if (tid == 0) cyc[4] = t1 - t0;
    // (d) compare -> ballot -> scalar branch, dependent
    double x = d; int cnt = 0;
    t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
        const unsigned long long m = __ballot(x * 0.5 > (double)it * 1e-30);
        if (__builtin_expect(m == 0ull, 0)) { x = x * 3.0; cnt++; }
        x = x + 1e-9; asm volatile("" : "+v"(x));
    }
    t1 = clock64(); if (tid == 0) cyc[5] = t1 - t0;
    // (e) 16 independent fmas per iteration
    double t[16]; for (int i = 0; i < 16; ++i) t[i] = x + i;
    t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { t[i] = fma(-a, b, t[i]); asm volatile("" : "+v"(t[i])); }
    }
    t1 = clock64(); if (tid == 0) cyc[6] = t1 - t0;
    // (f) wave max
    double w = t[0];
    t0 = clock64();
    for (int it = 0; it < ITERS; ++it) { w = wave_max_all(w + tid) * 0.5; asm volatile("" : "+v"(w)); }
    t1 = clock64(); if (tid == 0) cyc[7] = t1 - t0;
    // (g) LDS read latency alone (dependent address chain)
    int idx = tid & 255; buf[0][tid] = (double)((tid * 7 + 1) & 255); __syncthreads();
    t0 = clock64();
    for (int it = 0; it < ITERS; ++it) { idx = (int)buf[0][idx]; }
    t1 = clock64(); if (tid == 0) cyc[8] = t1 - t0;
    // (h) s_memtime pair overhead
    t0 = clock64();
    long long acc = 0;
    for (int it = 0; it < ITERS; ++it) { acc += clock64(); }
    t1 = clock64(); if (tid == 0) cyc[9] = t1 - t0;
    double s = v + a + d + x + w + cnt + idx + (double)(acc & 1); for (int i = 0; i < 16; ++i) s += t[i];
    out[tid] = s;
}
int main()
{
    double* out; long long* cyc; hipMalloc(&out, 256 * 8); hipMalloc(&cyc, 16 * 8);
    for (int rep = 0; rep < 2; ++rep) { k<<<1, 256>>>(out, cyc, 1.0); hipDeviceSynchronize(); }
    long long h[16]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const char* names[] = {"publish+barrier+read round trip (4 waves)", "ds_write -> waitcnt -> ds_read (no barrier)", "s_barrier alone (4 waves)", "dependent v_fma_f64", "rcp + 2 Newton (5 dependent ops) + add",
                           "cmp -> ballot -> scalar branch + add", "16 independent v_fma_f64", "wave max (4 DPP steps + 8 readlanes) + mul", "dependent LDS read (+ cvt)", "s_memtime in a loop"};
    for (int i = 0; i < 10; ++i) printf("%-50s %8.1f cycles / iteration\n", names[i], (double)h[i] / ITERS);
    return 0;
}

// microbenchmark: issue cost (cycles per instruction, one wavefront on its SIMD) of the ways a pivot-row entry can reach all lanes of a 16-lane row
//   (a) v_fmac_f64_dpp row_newbcast (what the 16 x 16 diagonal-block elimination uses), independent instructions
//   (b) plain v_fma_f64, independent   (c) 2 x v_readlane_b32 + v_fma_f64 with the scalar operand   (d) v_mov_b64_dpp row_newbcast
//   (e) dependent chain of v_fmac_f64_dpp (latency)   (f) dependent chain of v_fma_f64
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/dpp_rate.hip -o tools/micro/dpp_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define R16(x) x x x x x x x x x x x x x x x x
__global__ __launch_bounds__(64) void k(double* out, long long* cyc, double seed)
{
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    double t = seed * 0.5 + threadIdx.x, l = 1e-3;
    long long t0, t1;
    const int ITER = 64;
    t0 = clock64();
    for (int it = 0; it < ITER; ++it)
        asm volatile(R16("v_fmac_f64_dpp %0, %8, -%9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %8, -%9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %8, -%9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %8, -%9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %4, %8, -%9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %8, -%9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %6, %8, -%9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %8, -%9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(t), "v"(l));
    t1 = clock64(); if (threadIdx.x == 0) cyc[0] = t1 - t0;
    t0 = clock64();
    for (int it = 0; it < ITER; ++it)
        asm volatile(R16("v_fma_f64 %0, %8, -%9, %0\n\tv_fma_f64 %1, %8, -%9, %1\n\tv_fma_f64 %2, %8, -%9, %2\n\tv_fma_f64 %3, %8, -%9, %3\n\t"
                         "v_fma_f64 %4, %8, -%9, %4\n\tv_fma_f64 %5, %8, -%9, %5\n\tv_fma_f64 %6, %8, -%9, %6\n\tv_fma_f64 %7, %8, -%9, %7\n\t")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(t), "v"(l));
    t1 = clock64(); if (threadIdx.x == 0) cyc[1] = t1 - t0;
    int tlo = __double2loint(t), thi = __double2hiint(t);
    t0 = clock64();
    for (int it = 0; it < ITER; ++it)
        asm volatile(R16("v_readlane_b32 s20, %9, 1\n\tv_readlane_b32 s21, %10, 1\n\tv_readlane_b32 s22, %9, 2\n\tv_readlane_b32 s23, %10, 2\n\t"
                         "v_readlane_b32 s24, %9, 3\n\tv_readlane_b32 s25, %10, 3\n\tv_readlane_b32 s26, %9, 4\n\tv_readlane_b32 s27, %10, 4\n\t"
                         "v_fma_f64 %0, s[20:21], -%8, %0\n\tv_fma_f64 %1, s[22:23], -%8, %1\n\tv_fma_f64 %2, s[24:25], -%8, %2\n\tv_fma_f64 %3, s[26:27], -%8, %3\n\t"
                         "v_readlane_b32 s20, %9, 5\n\tv_readlane_b32 s21, %10, 5\n\tv_readlane_b32 s22, %9, 6\n\tv_readlane_b32 s23, %10, 6\n\t"
                         "v_readlane_b32 s24, %9, 7\n\tv_readlane_b32 s25, %10, 7\n\tv_readlane_b32 s26, %9, 8\n\tv_readlane_b32 s27, %10, 8\n\t"
                         "v_fma_f64 %4, s[20:21], -%8, %4\n\tv_fma_f64 %5, s[22:23], -%8, %5\n\tv_fma_f64 %6, s[24:25], -%8, %6\n\tv_fma_f64 %7, s[26:27], -%8, %7\n\t")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(l), "v"(tlo), "v"(thi)
                     : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    t1 = clock64(); if (threadIdx.x == 0) cyc[2] = t1 - t0;
    double d0, d1, d2, d3;
    t0 = clock64();
    for (int it = 0; it < ITER; ++it)
        asm volatile(R16("v_mov_b64_dpp %0, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b64_dpp %2, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %3, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b64_dpp %0, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %1, %4 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b64_dpp %2, %4 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %3, %4 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t")
                     : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(t));
    t1 = clock64(); if (threadIdx.x == 0) cyc[3] = t1 - t0;
    t0 = clock64();
    for (int it = 0; it < ITER; ++it)
        asm volatile(R16("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t")
                     : "+v"(a0) : "v"(l));
    t1 = clock64(); if (threadIdx.x == 0) cyc[4] = t1 - t0;
    t0 = clock64();
    for (int it = 0; it < ITER; ++it)
        asm volatile(R16("v_fma_f64 %0, %0, %1, %0\n\tv_fma_f64 %0, %0, %1, %0\n\tv_fma_f64 %0, %0, %1, %0\n\tv_fma_f64 %0, %0, %1, %0\n\t"
                         "v_fma_f64 %0, %0, %1, %0\n\tv_fma_f64 %0, %0, %1, %0\n\tv_fma_f64 %0, %0, %1, %0\n\tv_fma_f64 %0, %0, %1, %0\n\t")
                     : "+v"(a1) : "v"(l));
    t1 = clock64(); if (threadIdx.x == 0) cyc[5] = t1 - t0;
    double rr = seed + 2.0;
    t0 = clock64();
    for (int it = 0; it < ITER; ++it)
        asm volatile(R16("v_rcp_f64 %0, %0\n\ts_nop 0\n\tv_rcp_f64 %0, %0\n\ts_nop 0\n\tv_rcp_f64 %0, %0\n\ts_nop 0\n\tv_rcp_f64 %0, %0\n\ts_nop 0\n\tv_rcp_f64 %0, %0\n\ts_nop 0\n\tv_rcp_f64 %0, %0\n\ts_nop 0\n\tv_rcp_f64 %0, %0\n\ts_nop 0\n\tv_rcp_f64 %0, %0\n\ts_nop 0\n\t") : "+v"(rr));
    t1 = clock64(); if (threadIdx.x == 0) cyc[6] = t1 - t0;
    out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + d0 + d1 + d2 + d3 + rr;
}
int main()
{
    double* out; long long* cyc; hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8 * 8);
    for (int p = 0; p < 2; ++p) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, 1.25); hipDeviceSynchronize(); }
    long long h[8]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const double n = 64.0 * 16 * 8;
    printf("cycles per instruction (one wavefront): v_fmac_f64_dpp row_newbcast %.1f | v_fma_f64 %.1f | 2 x v_readlane_b32 + v_fma_f64(sgpr) %.1f per column | v_mov_b64_dpp %.1f | dependent v_fmac_f64_dpp (+ s_nop 1) %.1f | dependent v_fma_f64 %.1f | dependent v_rcp_f64 (+ s_nop 0) %.1f\n",
           h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[5] / n, h[6] / n);
    return 0;
}

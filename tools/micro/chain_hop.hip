// Hop latency of a chain of workgroups passing a 64-entry tagged message (the pipeline of the data-flow solve sweeps), per variant:
//   0  every wavefront polls all 64 entries; the 4 wavefronts of the producer store 16 entries each (lanes 0, 4, 8, ...)
//   1  as 0, but wavefront 0 stores all 64 entries (after a barrier)
//   2  wavefront 0 polls and hands the message on through LDS (one more barrier), 4 wavefronts store
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/chain_hop.hip -o tools/micro/chain_hop.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2d ld_tag(const v2d* p) { v2d r; asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory"); return r; }
__device__ __forceinline__ void st_tag(v2d* p, v2d v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }
template <int VAR>
__global__ __launch_bounds__(256) void hop(v2d* buf, double ep, unsigned long long* out, int filler)
{
    __shared__ double ys[4][64], acc[64];
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = tid >> 2, part = tid & 3;
    double y = 1.0;
    if (w > 0) {
        const v2d* p = buf + 64 * (size_t)(w - 1) * filler;
        v2d v;
        if (VAR == 2) {
            if (wave == 0) { do { v = ld_tag(p + lane); if (__all(v.y == ep)) break; __builtin_amdgcn_s_sleep(1); } while (true); ys[0][lane] = v.x; }
            __syncthreads();
        } else {
            do { v = ld_tag(p + lane); if (__all(v.y == ep)) break; __builtin_amdgcn_s_sleep(1); } while (true);
            ys[wave][lane] = v.x;
        }
        double t = 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) t += 0.001 * ys[VAR == 2 ? 0 : wave][part + 4 * u];
        t += __shfl_xor(t, 1); t += __shfl_xor(t, 2);
        if (part == 0) acc[row] = t;
        __syncthreads();
        y = acc[(row + 1) & 63] + 1.0;
    }
    v2d m; m.x = y; m.y = ep;
    v2d* q = buf + 64 * (size_t)w * filler;
    if (VAR == 1) { if (part == 0) acc[row] = y; __syncthreads(); if (wave == 0) { m.x = acc[lane]; st_tag(q + lane, m); } }
    else if (part == 0) st_tag(q + row, m);
    if (tid == 0) out[w] = wall_clock64();
}
int main()
{
    const int N = 64;
    v2d* buf; unsigned long long* out; unsigned long long h[N];
    hipMalloc(&buf, N * 64 * 16 * 64); hipMalloc(&out, N * 8);
    double ep = 0.0;
    for (int filler : {1, 64})
    for (int var = 0; var < 3; ++var) {
        for (int rep = 0; rep < 3; ++rep) {
            ep += 1.0;
            if (var == 0) hipLaunchKernelGGL(hop<0>, dim3(N), dim3(256), 0, 0, buf, ep, out, filler);
            if (var == 1) hipLaunchKernelGGL(hop<1>, dim3(N), dim3(256), 0, 0, buf, ep, out, filler);
            if (var == 2) hipLaunchKernelGGL(hop<2>, dim3(N), dim3(256), 0, 0, buf, ep, out, filler);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        printf("variant %d, messages %4d bytes apart: %.2f us per hop (chain of %d)\n", var, filler * 1024, (h[N - 1] - h[0]) * 0.01 / (N - 1), N);
    }
    return 0;
}

// Round-trip latency between two workgroups on different XCDs (MI355X), two protocols:
//   A  64 doubles + release fence + flag  /  flag poll + acquire fence + load      (what the chain sweeps did up to round 2)
//   B  64 x {value, tag} as 16-byte agent-coherent stores / polled with 16-byte agent-coherent loads (no fence, no flag)
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/pingpong.hip -o tools/micro/pingpong.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2d ld_tag(const v2d* p) { v2d r; asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory"); return r; }
__device__ __forceinline__ void st_tag(v2d* p, v2d v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }

__global__ __launch_bounds__(256) void pp_a(double* buf, int* flag, int n, unsigned long long* out, int other)
{
    const int me = blockIdx.x == 0 ? 0 : 1;
    if (blockIdx.x != 0 && (int)blockIdx.x != other) return;
    const int tid = threadIdx.x;
    __shared__ double ys[64];
    unsigned long long t0 = wall_clock64();
    double acc = 0.0;
    for (int i = 1; i <= n; ++i) {
        for (int side = 0; side < 2; ++side) {
            double* d = buf + 64 * side; int* f = flag + 32 * side;
            if (me == side) {            // produce
                if (tid < 64) d[tid] = acc + tid + i;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __syncthreads();
                if (tid == 0) __hip_atomic_store(f, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (tid == 0) while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != i) __builtin_amdgcn_s_sleep(1);
                __syncthreads();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                if (tid < 64) ys[tid] = d[tid];
                __syncthreads();
                acc = ys[(tid + 1) & 63];
            }
        }
    }
    if (tid == 0) { out[me] = wall_clock64() - t0; out[2 + me] = (unsigned long long)acc; }
}
template <int MODE>
__global__ __launch_bounds__(256) void pp_b(v2d* buf, int n, unsigned long long* out, int other)
{
    const int me = blockIdx.x == 0 ? 0 : 1;
    if (blockIdx.x != 0 && (int)blockIdx.x != other) return;
    const int tid = threadIdx.x, lane = tid & 63;
    __shared__ double ys[4][64];
    unsigned long long t0 = wall_clock64();
    double acc = 0.0;
    for (int i = 1; i <= n; ++i) {
        for (int side = 0; side < 2; ++side) {
            v2d* d = buf + 64 * side;
            if (me == side) {
                if (tid < 64) { v2d v; v.x = acc + tid + i; v.y = (double)i; st_tag(d + tid, v); }
            } else {             // every wavefront polls for itself: no barrier
                v2d v;
                int spins = 0;
                if (MODE == 1) {      // first entry only (one request per poll), then everything
                    do { v = ld_tag(d); if (__builtin_amdgcn_readfirstlane(__double2loint(v.y)) == __double2loint((double)i) && __builtin_amdgcn_readfirstlane(__double2hiint(v.y)) == __double2hiint((double)i)) break; __builtin_amdgcn_s_sleep(1); } while (++spins < 100000);
                    if (spins >= 100000) out[4] = i;
                }
                do { v = ld_tag(d + lane); if (__all(v.y == (double)i)) break; __builtin_amdgcn_s_sleep(1); } while (true);
                ys[tid >> 6][lane] = v.x;
                acc = ys[tid >> 6][(lane + 1) & 63];
            }
        }
    }
    if (tid == 0) { out[me] = wall_clock64() - t0; out[2 + me] = (unsigned long long)acc; }
}
int main()
{
    double* buf; int* flag; unsigned long long* out; v2d* tb;
    hipMalloc(&buf, 4096); hipMalloc(&flag, 4096); hipMalloc(&out, 64); hipMalloc(&tb, 4096);
    unsigned long long h[8];
    const int n = 2000;
    for (int other : {1, 8, 9, 64}) {
        hipMemset(buf, 0, 4096); hipMemset(flag, 0, 4096); hipMemset(tb, 0, 4096);
        hipLaunchKernelGGL(pp_a, dim3(other + 1), dim3(256), 0, 0, buf, flag, n, out, other);
        hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
        printf("A (fence + flag)   workgroups 0 and %2d: %.2f us per one-way message\n", other, h[0] * 0.01 / (2.0 * n));
        hipLaunchKernelGGL(pp_b<0>, dim3(other + 1), dim3(256), 0, 0, tb, n, out, other);
        hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
        printf("B (tagged 16-byte) workgroups 0 and %2d: %.2f us per one-way message\n", other, h[0] * 0.01 / (2.0 * n));
        hipMemset(tb, 0, 4096); hipMemset(out, 0, 64);
        hipLaunchKernelGGL(pp_b<1>, dim3(other + 1), dim3(256), 0, 0, tb, n, out, other);
        hipMemcpy(h, out, 40, hipMemcpyDeviceToHost);
        printf("C (first entry polled) workgroups 0 and %2d: %.2f us per one-way message (poll gave up at message %llu)\n", other, h[0] * 0.01 / (2.0 * n), h[4]);
    }
    return 0;
}

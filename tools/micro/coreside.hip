// Can a small workgroup share a CU with a workgroup of a register- and LDS-heavy kernel launched on ANOTHER stream (MI355X)?
// The chain-group kernel (k_grp_fused: 256 threads, 386 VGPRs, 110-143 KB of LDS) pins one CU per workgroup for the ~160 us of its latency
// chain; the trailing update of the previous group runs next to it on the second stream.  HOST<NV> below stands for it: 256 threads keeping
// 2 NV VGPRs live, `lds` bytes of dynamic LDS, spinning for `dur` us -- one workgroup per CU.  GUEST<NV, NT> is launched on a second stream a
// few us later; every guest workgroup records when it started (wall clock relative to the host launch's first workgroup) and on which CU.
// If the guests start while the hosts still spin they share CUs; if they start when the hosts leave they do not.
// build: hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage tools/micro/coreside.hip -o tools/micro/coreside.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NV>
__global__ __launch_bounds__(256) void host_k(const double* in, double* out, unsigned long long* t_first, long long dur_ticks)
{
    extern __shared__ double sm[];
    double r[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) r[i] = in[(threadIdx.x + i) & 255];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) atomicMin(t_first, t0);
    sm[threadIdx.x] = r[0];
    while ((long long)(wall_clock64() - t0) < dur_ticks) {
#pragma unroll
        for (int i = 0; i < NV; ++i) r[i] = r[i] * 1.0000001 + 1e-9;
        __builtin_amdgcn_s_sleep(8);
    }
    double s = sm[(threadIdx.x + 1) & 255];
#pragma unroll
    for (int i = 0; i < NV; ++i) s += r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV, int NT>
__global__ __launch_bounds__(NT) void guest_k(const double* in, double* out, const unsigned long long* t_first, long long* started, int* where, long long dur_ticks)
{
    extern __shared__ double sm[];
    double r[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) r[i] = in[(threadIdx.x + i) & 255];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        started[blockIdx.x] = (long long)(t0 - *t_first);
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        where[blockIdx.x] = (int)(((xcc & 0xf) << 16) | (hw & 0xffff));
    }
    sm[threadIdx.x & 255] = r[0];
    while ((long long)(wall_clock64() - t0) < dur_ticks) {
#pragma unroll
        for (int i = 0; i < NV; ++i) r[i] = r[i] * 1.0000001 + 1e-9;
    }
    double s = sm[(threadIdx.x + 1) & 255];
#pragma unroll
    for (int i = 0; i < NV; ++i) s += r[i];
    out[blockIdx.x * NT + threadIdx.x] = s;
}

template <int HV, int GV, int GNT>
int run(const char* name, size_t host_lds, size_t guest_lds, int nhost, int nguest)
{
    double *in, *out; unsigned long long* tf; long long* st; int* wh;
    CK(hipMalloc(&in, 256 * 8)); CK(hipMalloc(&out, (size_t)4096 * 1024 * 8)); CK(hipMalloc(&tf, 8)); CK(hipMalloc(&st, 4096 * 8)); CK(hipMalloc(&wh, 4096 * 4));
    CK(hipMemset(in, 0, 256 * 8));
    CK(hipFuncSetAttribute((const void*)host_k<HV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)guest_k<GV, GNT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const long long host_ticks = 30000, guest_ticks = 500;      // wall clock = 100 MHz: 300 us, 5 us
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(tf, 0xff, 8)); CK(hipMemset(st, 0, 4096 * 8)); CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(host_k<HV>, dim3(nhost), dim3(256), host_lds, s1, in, out, tf, host_ticks);
        for (volatile int spin = 0; spin < 200000; ++spin) {}      // (the hosts are in place before the guests are sent)
        hipLaunchKernelGGL((guest_k<GV, GNT>), dim3(nguest), dim3(GNT), guest_lds, s2, in, out + 2048 * 256, tf, st, wh, guest_ticks);
        CK(hipDeviceSynchronize());
    }
    std::vector<long long> h(nguest); std::vector<int> w(nguest);
    CK(hipMemcpy(h.data(), st, nguest * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(w.data(), wh, nguest * 4, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    int early = 0; for (long long x : h) if (x < host_ticks - 1000) ++early;
    printf("%-58s host lds %6zu guest lds %6zu: %4d of %4d guests started while the hosts were spinning; start (us) min %.1f median %.1f max %.1f\n", name, host_lds, guest_lds, early, nguest,
           h[0] / 100.0, h[nguest / 2] / 100.0, h[nguest - 1] / 100.0);
    hipFree(in); hipFree(out); hipFree(tf); hipFree(st); hipFree(wh); hipStreamDestroy(s1); hipStreamDestroy(s2);
    return 0;
}

int main()
{
    // hosts: 140 doubles live = 256 VGPRs + 128 AGPRs (the resource remarks of the build); 256 workgroups = one per CU
    run<140, 40, 256>("host 380+ VGPR x 256 WGs, guest ~90 VGPR 256 thr", 110 * 1024, 0, 256, 1024);
    run<140, 40, 256>("host 380+ VGPR x 256 WGs, guest ~90 VGPR 256 thr", 110 * 1024, 35 * 1024, 256, 1024);
    run<140, 40, 256>("host 380+ VGPR x 256 WGs, guest ~90 VGPR 256 thr", 143 * 1024, 0, 256, 1024);
    run<140, 40, 256>("host 380+ VGPR x 256 WGs, guest ~90 VGPR 256 thr", 143 * 1024, 35 * 1024, 256, 1024);
    run<140, 54, 256>("host 380+ VGPR x 256 WGs, guest ~118 VGPR 256 thr", 110 * 1024, 0, 256, 1024);
    run<140, 24, 256>("host 380+ VGPR x 256 WGs, guest ~56 VGPR 256 thr", 110 * 1024, 0, 256, 1024);
    run<140, 40, 1024>("host 380+ VGPR x 256 WGs, guest ~90 VGPR 1024 thr", 80 * 1024, 66 * 1024, 256, 512);
    run<120, 40, 1024>("host 250 VGPR x 256 WGs, guest ~90 VGPR 1024 thr", 77 * 1024, 66 * 1024, 256, 512);
    run<56, 40, 1024>("host 120 VGPR x 256 WGs, guest ~90 VGPR 1024 thr", 77 * 1024, 66 * 1024, 256, 512);
    run<56, 40, 1024>("host 120 VGPR x 512 WGs, guest ~90 VGPR 1024 thr", 77 * 1024, 0, 512, 512);
    // 188 hosts (the four-chain level): the guests have 68 CUs to themselves
    run<140, 40, 256>("host 380+ VGPR x 188 WGs, guest ~90 VGPR 256 thr", 110 * 1024, 35 * 1024, 188, 2048);
    return 0;
}

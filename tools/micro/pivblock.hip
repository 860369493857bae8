// microbenchmark: the blocked a-posteriori LDL^T of ONE 64 x 64 pivot block (ldlt_blocked_static, the serial spine of every chain link at the top of the
// tree) in isolation -- one workgroup of 256 threads on an otherwise idle chip, the block in LDS -- with the phase stamps the kernel already carries:
// per 16-column sub-block  A (diagonal block + inverse on wavefront 0) | B (rows below, MFMA) | C (trailing tiles, MFMA), in shader cycles.
// In k_grp_fused the same code measured 5 400 | 2 200 | 3 000-5 000 cycles per sub-block (profiles/r04b_synth_1e6_pivot_block_phases.txt); the DPP
// arithmetic of A is ~1 400.  build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/pivblock.hip ipopt_amd/lib/symbolic.o ipopt_amd/lib/matching_scaling.o -o tools/micro/pivblock.bin
#include "../../ipopt_amd/csrc/numeric.hip"
namespace mi355x {
__global__ __launch_bounds__(256) void k_probe(const double* A, unsigned long long* out, double* Lout, int k, int reps, int variant)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, ld = k | 1;
    double* Lb = reinterpret_cast<double*>(smem_raw);
    double* dinv_s = Lb + (size_t)ld * k + 64;
    double* Wp = dinv_s + 64;
    double* Isb = Wp + 16 * 65 * 4;
    int* shflag = reinterpret_cast<int*>(Isb + 4 * 272);
    unsigned long long ts[28];
    for (int q = 0; q < 28; ++q) ts[q] = 0ull;
    unsigned long long acc[12];
    for (int q = 0; q < 12; ++q) acc[q] = 0ull;
    unsigned long long total = 0;
    for (int r = 0; r < reps; ++r) {
        for (int idx = tid; idx < k * k; idx += 256) { const int i = idx % k, c = idx / k; Lb[i + c * ld] = A[i + c * k]; }
        __syncthreads();
        int nneg = 0;
        if (variant == 1) { asm volatile("s_icache_inv\n\ts_nop 7\n\ts_nop 7" ::: "memory"); __syncthreads(); }      // (every pass starts with a cold instruction cache, as in k_grp_fused where a workgroup runs this code ONCE)
        const unsigned long long t0 = clock64();
        const bool ok = ldlt_blocked_static(Lb, ld, k, Wp, dinv_s, Isb, shflag, 1e-300, 1e8, nneg, nullptr, ts);
        const unsigned long long t1 = clock64();
        __syncthreads();
        if (r > 0) {   // (the first pass is cold: instruction cache)
            total += t1 - t0;
            acc[0] += ts[16] - t0; acc[1] += ts[17] - ts[16]; acc[2] += ts[18] - ts[17];
            acc[3] += ts[19] - ts[18]; acc[4] += ts[20] - ts[19]; acc[5] += ts[21] - ts[20];
            acc[6] += ts[22] - ts[21]; acc[7] += ts[23] - ts[22]; acc[8] += ts[24] - ts[23];
            acc[9] += ts[25] - ts[24];
        } else if (tid == 0) { out[20] = t1 - t0; out[21] = ok ? 1 : 0; out[22] = nneg; }
    }
    if (tid == 0) { out[0] = total; for (int q = 0; q < 10; ++q) out[1 + q] = acc[q]; }
    for (int idx = tid; idx < k * k; idx += 256) { const int i = idx % k, c = idx / k; Lout[i + c * k] = Lb[i + c * ld]; }
}
}
int main(int argc, char** argv)
{
    const int k = argc > 1 ? atoi(argv[1]) : 64, reps = 21, variant = argc > 2 ? atoi(argv[2]) : 0;
    std::vector<double> A((size_t)k * k);
    for (int c = 0; c < k; ++c) for (int i = 0; i < k; ++i) A[i + (size_t)c * k] = (i == c) ? (4.0 + 0.01 * i) * ((i % 3) ? 1.0 : -1.0) : 0.3 / (1.0 + abs(i - c));
    double *dA, *dL; unsigned long long* dout;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dL, A.size() * 8); hipMalloc(&dout, 64 * 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)mi355x::k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(mi355x::k_probe, dim3(1), dim3(256), 120 * 1024, 0, dA, dout, dL, k, reps, variant);
        hipDeviceSynchronize();
    }
    unsigned long long out[64]; hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost);
    const double n = reps - 1;
    { std::vector<double> L(A.size()); hipMemcpy(L.data(), dL, L.size() * 8, hipMemcpyDeviceToHost); unsigned long long h = 1469598103934665603ull;
      for (double v : L) { unsigned long long b; memcpy(&b, &v, 8); h = (h ^ b) * 1099511628211ull; } printf("  checksum of the factor's bits: %016llx\n", h); }
    printf("k = %d: blocked LDL^T of the pivot block, warm: %.0f cycles per block (cold first pass %llu), accepted %llu, negative pivots %llu\n", k, out[0] / n, out[20], out[21], out[22]);
    printf("  sub-block 0: A %.0f | B %.0f | C %.0f\n  sub-block 1: A %.0f | B %.0f | C %.0f\n  sub-block 2: A %.0f | B %.0f | C %.0f\n  sub-block 3: A %.0f\n",
           out[1] / n, out[2] / n, out[3] / n, out[4] / n, out[5] / n, out[6] / n, out[7] / n, out[8] / n, out[9] / n, out[10] / n);
    return 0;
}

// How long do hipMalloc and hipMemset of a pool of `GiB` take on this box (the one-time cost of "every contribution block resident": MBndryCntrl_3D 78 = 74 GiB)?
// build: hipcc -O2 --offload-arch=gfx950 tools/micro/malloc_time.hip -o tools/micro/malloc_time.bin ; usage: malloc_time.bin [GiB ...]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    (void)hipFree(nullptr);
    for (int a = 1; a < argc || a == 1; ++a) {
        const double gib = a < argc ? atof(argv[a]) : 74.0;
        const size_t bytes = (size_t)(gib * 1024.0 * 1024.0 * 1024.0);
        void* p = nullptr;
        double t0 = now();
        if (hipMalloc(&p, bytes) != hipSuccess) { printf("%.1f GiB: hipMalloc failed\n", gib); continue; }
        double t1 = now();
        (void)hipMemset(p, 0, bytes); (void)hipDeviceSynchronize();
        double t2 = now();
        (void)hipMemset(p, 0, bytes); (void)hipDeviceSynchronize();
        double t3 = now();
        (void)hipFree(p);
        double t4 = now();
        printf("%6.1f GiB: hipMalloc %.3f s, first hipMemset %.3f s, second hipMemset %.3f s, hipFree %.3f s\n", gib, t1 - t0, t2 - t1, t3 - t2, t4 - t3);
        if (a >= argc) break;
    }
    // the same 74 GiB as ten pieces, and as one virtual range backed by ten physical pieces (hipMemAddressReserve / hipMemCreate / hipMemMap)
    {
        const size_t piece = (size_t)(7.4 * 1024.0 * 1024.0 * 1024.0);
        void* q[10]; double t0 = now();
        for (int i = 0; i < 10; ++i) if (hipMalloc(&q[i], piece) != hipSuccess) q[i] = nullptr;
        double t1 = now();
        for (int i = 0; i < 10; ++i) if (q[i]) (void)hipFree(q[i]);
        printf("10 x 7.4 GiB: hipMalloc %.3f s in all\n", t1 - t0);
    }
    {
        hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        size_t gran = 0; (void)hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
        if (gran == 0) gran = 2u << 20;
        size_t piece = (size_t)(7.4 * 1024.0 * 1024.0 * 1024.0); piece = (piece + gran - 1) / gran * gran;
        void* base = nullptr; double t0 = now();
        if (hipMemAddressReserve(&base, 10 * piece, gran, nullptr, 0) == hipSuccess) {
            hipMemGenericAllocationHandle_t h[10]; int ok = 1;
            for (int i = 0; i < 10 && ok; ++i) {
                ok = hipMemCreate(&h[i], piece, &prop, 0) == hipSuccess && hipMemMap((char*)base + i * piece, piece, 0, h[i], 0) == hipSuccess;
            }
            hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
            if (ok) ok = hipMemSetAccess(base, 10 * piece, &acc, 1) == hipSuccess;
            double t1 = now();
            if (ok) { (void)hipMemset(base, 0, 10 * piece); (void)hipDeviceSynchronize(); }
            double t2 = now();
            printf("one virtual range of 10 x 7.4 GiB physical pieces (granularity %zu): reserve + create + map + access %.3f s (%s), hipMemset %.3f s\n", gran, t1 - t0, ok ? "ok" : "FAILED", t2 - t1);
        } else printf("hipMemAddressReserve failed\n");
    }
    return 0;
}

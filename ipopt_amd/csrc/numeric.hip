// numeric.hip -- hand-written HIP (gfx950 / CDNA4) numeric engine of the MI355X KKT solver:
// value gather + equilibration, level-scheduled multifrontal LDL^T with Bunch-Kaufman
// pivoting inside the fully-summed block of every front, inertia count, and
// level-scheduled forward / diagonal / backward solves.
//
// This is the arithmetic the reference delegates to MUMPS / HSL / PARDISO behind
// SparseSymLinearSolverInterface::MultiSolve (IpSparseSymLinearSolverInterface.hpp:190;
// e.g. IpMumpsSolverInterface.cpp:448-583, IpMa97SolverInterface.cpp:611-820); the value
// gather replaces TripletToCSRConverter::ConvertValues (IpTripletToCSRConverter.cpp:337-372).
//
// Design:
//   * wave = 64 lanes; small fronts (order <= 32) get ONE wavefront each, fronts up to order
//     128 one 256-thread workgroup; the whole front lives in LDS (<= 134 KiB of the 160 KiB/CU)
//     from assembly to the write-back of L and of the contribution block, so HBM sees each
//     A value, each child contribution block and each L entry exactly once.
//   * column-major fronts with an ODD leading dimension: lanes walk rows => consecutive 8-byte
//     LDS words (ds_read_b64 is conflict-free); the occasional row walk strides by an odd
//     number of 8-byte banks.
//   * pivot search / column maxima are wavefront shuffle reductions (DPP), one LDS hop across
//     the 4 waves of a workgroup.
//   * everything is launched on one HIP stream, one launch per (tree level, front class);
//     the sequences are captured into hipGraphs and replayed (launch-bound regime).
//   * larger fronts take the blocked global-memory path (panel kernels + v_mfma_f64_16x16x4
//     trailing updates), see bigfront section.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types only: librccl.so is dlopen()ed when a communicator is requested (multi-GPU), never linked
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <chrono>
#include <vector>
#include <thread>
#include <map>
#include <string>
#include <algorithm>
#include "numeric.h"
#include "env_knobs.h"
#include "matching_scaling.h"

namespace mi355x {

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    err_ = std::string(#call) + ": " + hipGetErrorString(e_); return false; } } while (0)

// kernel kinds for the profiling entry point (order = include/mi355x_kkt.h MI355X_KKT_KERNEL_*)
enum KernelKind { KK_GATHER_SCALE = 0, KK_FRONT_WAVE, KK_FRONT_LDS64, KK_FRONT_LDS128, KK_BIG_ASSEMBLE, KK_BIG_DIAG, KK_BIG_TRSM,
                  KK_BIG_SCHUR, KK_STATS, KK_SOLVE_PERM, KK_FWD_WAVE, KK_FWD_LDS, KK_FWD_BIG, KK_BWD_WAVE, KK_BWD_LDS, KK_BWD_BIG, KK_FWD_BIG_UPD, KK_BWD_BIG_DOT, KK_COUNT };
#define DBGSTAMP(slot) do { if (V.dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { V.dbg[2 * (slot)] = clock64(); V.dbg[2 * (slot) + 1] = wall_clock64(); } } while (0)
#define DBGT(i) do { if (V.dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) V.dbg[16 + (i)] = clock64(); } while (0)
// (MI355X_KKT_TRACE=launches: development -- name the launch whose configuration the runtime refuses instead of the "invalid configuration argument" the next HIPCHK would report)
#include "env_knobs.h"
static const bool g_launch_check = mi355x::knob_trace("launches");
#define LAUNCH_VERIFY(what) do { if (g_launch_check) { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) fprintf(stderr, "[mi355x_kkt] launch refused (%s): %s  @ line %d\n", hipGetErrorString(e_), what, __LINE__); } } while (0)
#define LAUNCH(kind, ...) do { prof_begin(kind); hipLaunchKernelGGL(__VA_ARGS__); prof_end(); LAUNCH_VERIFY(#__VA_ARGS__); } while (0)
#define LAUNCH_ON(kind, strm, ...) do { prof_begin(kind, strm); hipLaunchKernelGGL(__VA_ARGS__); prof_end(strm); LAUNCH_VERIFY(#__VA_ARGS__); } while (0)      // (a launch on the look-ahead stream: its events are recorded there)

static constexpr double BK_ALPHA = 0.6403882032022076;   // (1+sqrt(17))/8
static constexpr double BK_ALPHA0 = 0.1;                 // a diagonal within this factor of its whole remaining column is taken as it comes (no partner search)
static constexpr double PIV_PERT = 1e-10;                // replacement magnitude for a zero pivot
static constexpr int ISG_STRIDE = 8 * 272;               // doubles per big front in DevView::isg
static constexpr double ZERO_REL = 1e-14;                // zero-pivot test relative to the largest entry assembled into the pivot's column

#include "device_view.hip.inc"
#include "kernels_values.hip.inc"
#include "kernels_fronts.hip.inc"
#include "kernels_solve.hip.inc"
#include "kernels_big.hip.inc"
#include "kernels_pd_multigpu.hip.inc"
#include "kernels_match.hip.inc"
// ------------------------------------------------------------------------------------------------
// host-side orchestration
// ------------------------------------------------------------------------------------------------
// every public entry point selects the handle's device and restores the caller's on exit: a host application (or another
// handle on another GPU / thread) may have switched the current device since setup()
struct DeviceGuard {
    int prev = -1; bool sw = false;
    explicit DeviceGuard(int dev) { if (dev >= 0 && hipGetDevice(&prev) == hipSuccess && prev != dev) sw = (hipSetDevice(dev) == hipSuccess); }
    ~DeviceGuard() { if (sw) (void)hipSetDevice(prev); }
};

class NumericImpl {
public:
    std::string err_;
    int dev = -1;            // HIP device ordinal of this handle
    const Symbolic* S = nullptr;
    NumericOptions opt;
    bool have_device = false, ready = false, have_values = false;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double factor_ms = 0, solve_ms = 0;
    double* h_vals = nullptr;     // pinned
    int* h_stats = nullptr;       // pinned, 4 ints
    std::vector<void*> allocs;
    DevView V{};
    int* d_stats = nullptr;
    double* d_rhs = nullptr; size_t d_rhs_cap = 0;
    hipGraphExec_t g_factor = nullptr, g_solve = nullptr;
    // Ruiz factors kept across a refactorisation in which ONLY the shifts of the assembled segments changed (an inertia-correction retry of the
    // device routes: new delta_x / delta_c, nothing uploaded, IpPDFullSpaceSolver.cpp:486-640): gather + apply instead of gather + row view + 4 sweeps
    // + apply (synth_1e6: 0.93 -> ~0.25 ms of a 16.7 ms factorisation, LukVlE1 10^6: 0.23 -> 0.07 of 0.69).  The reference's MA97 adapter keeps
    // its scaling factors until asked too (IpMa97SolverInterface.cpp:725-771,824-840); they are recomputed here on new W / J / Sigma values, on
    // IncreaseQuality (set_pivtol), after a structure edit and when the scaling mode changes.  Separate graph replays (g_factor_keep*).
    bool scale_valid = false, keep_scale_now = false, asm_dirty = true;
    double asm_prev_scale[16] = {0};
    hipGraphExec_t g_factor_keep = nullptr, g_factor_full_keep = nullptr;
    void destroy_factor_graphs() {
        for (hipGraphExec_t* g : {&g_factor, &g_factor_full, &g_factor_keep, &g_factor_full_keep}) if (*g) { (void)hipGraphExecDestroy(*g); *g = nullptr; }
    }
    bool scale_identity = true;
    double* d_user_scale = nullptr;      // caller-supplied scaling factors, original numbering (scaling mode 2)
    // scaling mode at run time: 0 none, 1 Ruiz on device, 2 the caller's factors (MA97 semantics, IpMa97SolverInterface.cpp:641-678)
    bool set_scaling(int mode, const double* user) {
        DeviceGuard guard(dev);
        if (!ready) { err_ = "set_scaling: solver not set up"; return false; }
        if (mode < 0 || mode > 6 || (mode == 2 && !user)) { err_ = "set_scaling: mode 0 (none), 1 (ruiz), 2 (user factors, non-null), 3 (matching, host), 4 (matching, host, reused), 5 (matching, device) or 6 (matching, device, reused)"; return false; }
        match_valid = false;                                  // (mode 4: the factors of an earlier selection do not survive a new one)
        if (mode == 2) {
            if (!d_user_scale) { HIPCHK(hipMalloc((void**)&d_user_scale, std::max<size_t>(S->n, 1) * sizeof(double))); allocs.push_back(d_user_scale); }
            HIPCHK(hipMemcpyAsync(d_user_scale, user, (size_t)S->n * sizeof(double), hipMemcpyHostToDevice, stream));
            HIPCHK(hipStreamSynchronize(stream));      // `user` is the caller's pageable memory
        }
        if (mode >= 3 && !d_user_scale) { HIPCHK(hipMalloc((void**)&d_user_scale, std::max<size_t>(S->n, 1) * sizeof(double))); allocs.push_back(d_user_scale); }
        if (mode != opt.scaling) destroy_factor_graphs();      // the captured sequences differ
        scale_valid = false;
        opt.scaling = mode;
        return true;
    }
    // scaling mode 3: maximum-product matching scaling (MC64-style, matching_scaling.cpp) of the values now in V.tvals.  Host
    // algorithm, like the analysis: gather on the device, one D2H of the nnz(A) summed values, the matching, one H2D of n
    // factors -- which the factorisation then applies exactly like caller-supplied ones.
    // scaling mode 4 = mode 3 computed ONCE and kept: the factors of the first factorisation serve the following ones until the caller asks for
    // better quality (IncreaseQuality -> invalidate_matching) -- MA97's "...-reuse" switches (IpMa97SolverInterface.cpp:725-771,824-840)
    bool match_valid = false;
    void invalidate_matching() { match_valid = false; }
    bool compute_matching_scaling() {
        const Symbolic& Sy = *S;
        hipLaunchKernelGGL(k_gather_values, dim3(grid1d(Sy.nnz_a)), dim3(256), 0, stream, V);
        std::vector<double> av(std::max(Sy.nnz_a, 1));
        HIPCHK(hipMemcpyAsync(av.data(), V.aval, (size_t)Sy.nnz_a * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        std::vector<double> absval(Sy.rslot_idx.size()), sp(std::max(Sy.n, 1)), so(std::max(Sy.n, 1));
        for (size_t p = 0; p < absval.size(); ++p) absval[p] = std::fabs(av[Sy.rslot_idx[p]]);
        int unmatched = 0;
        if (!matching_scaling(Sy.n, Sy.rslot_ptr.data(), Sy.rslot_col.data(), absval.data(), sp.data(), &unmatched)) { err_ = "matching scaling failed"; return false; }
        for (int i = 0; i < Sy.n; ++i) so[Sy.perm[i]] = sp[i];
        match_unmatched = unmatched; match_rounds = 0; match_ms = 0.f;
        if (opt.verbose) fprintf(stderr, "[mi355x_kkt] matching scaling: %d unmatched columns\n", unmatched);
        HIPCHK(hipMemcpyAsync(d_user_scale, so.data(), (size_t)Sy.n * sizeof(double), hipMemcpyHostToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream));          // `so` is about to go out of scope
        return true;
    }
    // scaling modes 5 / 6: the same job on the DEVICE (kernels_match.hip.inc: Jacobi auction over the symmetric row view; specification
    // tests/support/auction_spec.py).  Nothing crosses PCIe but the free-column counts that steer the rounds.  Mode 6 = computed once and kept like mode 4.
    std::vector<void*> match_allocs;
    MatchView MV{};
    int match_unmatched = 0, match_rounds = 0, match_launches = 0;
    float match_ms = 0.f;
    hipEvent_t match_e0 = nullptr, match_e1 = nullptr;
    void match_free() { for (void* p : match_allocs) (void)hipFree(p); match_allocs.clear(); MV = MatchView{};
                        if (match_e0) { (void)hipEventDestroy(match_e0); match_e0 = nullptr; } if (match_e1) { (void)hipEventDestroy(match_e1); match_e1 = nullptr; } }
    template <class T> bool match_alloc(T** d, size_t count) { T* p = nullptr; HIPCHK(hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T))); match_allocs.push_back(p); *d = p; return true; }
    bool compute_matching_device() {
        const Symbolic& Sy = *S; const int n = Sy.n;
        if (MV.n != n || MV.len != V.rslot_len || match_allocs.empty()) {
            match_free();
            const size_t len = (size_t)V.rslot_len;
            if (!match_alloc(&MV.b, len) || !match_alloc(&MV.logcmax, n) || !match_alloc(&MV.price, n) || !match_alloc(&MV.bidval, n) || !match_alloc(&MV.owner, n) ||
                !match_alloc(&MV.mrow, n) || !match_alloc(&MV.wcol, n) || !match_alloc(&MV.bidrow, n) || !match_alloc(&MV.bidkey, n) || !match_alloc(&MV.list0, n) ||
                !match_alloc(&MV.list1, n) || !match_alloc(&MV.cnt, 8)) return false;
            MV.n = n; MV.len = V.rslot_len;
        }
        MV.ptr = V.rslot_ptr; MV.col = V.rslot_col; MV.arv = V.arv; MV.perm = V.perm;       // (a structure edit for delayed pivots relabels the row view)
        if (!match_e0) { HIPCHK(hipEventCreate(&match_e0)); HIPCHK(hipEventCreate(&match_e1)); }      // (members: an error return below must not leak them)
        hipEvent_t e0 = match_e0, e1 = match_e1;
        HIPCHK(hipEventRecord(e0, stream));
        hipLaunchKernelGGL(k_gather_values, dim3(grid1d(Sy.nnz_a)), dim3(256), 0, stream, V);
        hipLaunchKernelGGL(k_abs_rowview, dim3(grid1d(V.rslot_len)), dim3(256), 0, stream, V);
        hipLaunchKernelGGL(k_match_init, dim3(grid1d(8ll * n)), dim3(256), 0, stream, MV);
        int hc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        auto read_cnt = [&]() -> bool { HIPCHK(hipMemcpyAsync(hc, MV.cnt, sizeof hc, hipMemcpyDeviceToHost, stream)); HIPCHK(hipStreamSynchronize(stream)); return true; };
        const double eps_final = 1.0 / 64;
        const int phase_rounds = 256, final_rounds = 8192, batch = 8;
        match_rounds = 0; match_launches = 3;
        for (double eps = 0.25; ; eps = std::max(eps / 4.0, eps_final)) {
            const bool last = eps <= eps_final;
            const int cap = last ? final_rounds : phase_rounds;
            HIPCHK(hipMemsetAsync(MV.cnt, 0, 8 * sizeof(int), stream));
            hipLaunchKernelGGL(k_match_phase, dim3(grid1d(8ll * n)), dim3(256), 0, stream, MV, eps);
            ++match_launches;
            if (!read_cnt()) return false;
            int cur = 0, nfree = hc[0], rounds = 0;
            while (nfree > 0 && rounds < cap) {
                if (nfree <= MATCH_TAIL) {
                    hipLaunchKernelGGL(k_match_tail, dim3(1), dim3(1024), 0, stream, MV, cur, eps, cap - rounds);
                    ++match_launches;
                    if (!read_cnt()) return false;
                    cur = hc[4]; nfree = hc[cur]; rounds += hc[3];
                    break;                          // (the tail kernel walks until no column is free or the limit)
                }
                const int nb = std::min(batch, cap - rounds);
                for (int r = 0; r < nb; ++r) {
                    hipLaunchKernelGGL(k_match_bid, dim3(grid1d(8ll * nfree)), dim3(256), 0, stream, MV, cur, eps);
                    hipLaunchKernelGGL(k_match_win, dim3(grid1d(nfree)), dim3(256), 0, stream, MV, cur);
                    hipLaunchKernelGGL(k_match_apply, dim3(grid1d(nfree)), dim3(256), 0, stream, MV, cur);
                    cur ^= 1;
                }
                match_launches += 3 * nb; rounds += nb;
                if (!read_cnt()) return false;
                nfree = hc[cur];
            }
            match_rounds += rounds;
            if (last) break;
        }
        HIPCHK(hipMemsetAsync(MV.cnt, 0, 8 * sizeof(int), stream));
        hipLaunchKernelGGL(k_match_final, dim3(grid1d(8ll * n)), dim3(256), 0, stream, MV, d_user_scale);
        ++match_launches;
        HIPCHK(hipEventRecord(e1, stream));
        if (!read_cnt()) return false;
        match_unmatched = hc[2];
        (void)hipEventElapsedTime(&match_ms, e0, e1);
        if (opt.verbose) fprintf(stderr, "[mi355x_kkt] matching scaling on the device: %d rounds, %d launches, %.3f ms, %d unmatched columns\n", match_rounds, match_launches, match_ms, match_unmatched);
        return true;
    }
    bool want_matching() const { return opt.scaling == 3 || opt.scaling == 5 || ((opt.scaling == 4 || opt.scaling == 6) && !match_valid); }
    bool run_matching() { return opt.scaling >= 5 ? compute_matching_device() : compute_matching_scaling(); }
    // the symmetric scaling of the last factorisation, original numbering (what MA97 writes into scale[])
    bool get_scaling(double* out) {
        DeviceGuard guard(dev);
        if (!ready) { err_ = "get_scaling: solver not set up"; return false; }
        std::vector<double> sp(S->n);
        if (S->n > 0) HIPCHK(hipMemcpy(sp.data(), V.scale, (size_t)S->n * sizeof(double), hipMemcpyDeviceToHost));
        for (int i = 0; i < S->n; ++i) out[S->perm[i]] = sp[i];
        return true;
    }
    std::vector<int> big_maxm, big_maxk, big_tiles, big_tiles64, big_last0, big_last1;
    std::vector<int> lv_narrow_tiles;   // > 0: every big front of the level is a chain link with a narrow update; the 64 x 64 tiles of the largest one
    std::vector<char> lv_asm_skip;      // every big front of the level is a pure in-place chain link: no assembly launch at all
    // sync-free chain sweeps: runs of consecutive levels made of pure chain links (single-GPU schedule, per-link solves)
    size_t strace_n = 0; struct TraceDesc { int chain, w, nlinks, tail; }; std::vector<TraceDesc> strace_desc;
    std::vector<char> in_seg;         // fronts handled by the data-flow sweeps
    struct ChainSeg { int lv0, lv1, desc0, ndesc, nwg_f, nwg_b, maxtail, wgf0, wgb0; };
    std::vector<ChainSeg> chain_segs; std::vector<int> seg_at_lv0, seg_at_lv1;      // level -> segment index (or -1)
    bool pair_solve = true; std::vector<int> wave_kmax, wave_mmax, wave_mmin;   // solves of the order <= 32 fronts: two fronts per wavefront (k_fwd_pair / k_bwd_pair); largest pivot count per level
    bool fuse_dt = true;                               // pivot block + panel solve in one launch where a level has few fronts
    int fuse_dt_maxwg = 448;                           // ... few = this many workgroups (pivot blocks + 64-row panel blocks) at most
    bool chain_solve = true; int chain_maxc = 128;       // only where few chains run side by side (the latency-bound top of the tree)
    std::vector<char> lv_allsolo;       // every big solve unit of the level is one link with nothing to gather (fused forward kernel)
    std::vector<int> big_split, part_mm[2], part_kk[2], part_tiles[2];   // single-GPU schedule: BIG buckets split at 1024 rows
    // look-ahead of the group-end trailing updates (single-GPU schedule): per level the grids of the two parts, second stream
    std::vector<int> la_tiles1, la_tiles2; std::vector<char> la_full;     // la_full: the level has a full (group-last) update
    // grouped schedule (single GPU): per level the chain groups whose FIRST link sits there (entries = FrontMeta of the LAST link, sorted by
    // order, split at 1024 rows like the BIG buckets), launch geometry, look-ahead tiles
    bool grouped = false;
    struct GrpSched { std::vector<int> g0, g1, split, nrb, tiles64, tiles, la1, la2, p1t, la3, nsplit; std::vector<hipEvent_t> evA, evB; };      // p1t: 64 x 64 tiles of a split front's part 1, la3: tiles of the fronts not split, nsplit: split fronts (among the large ones)
    GrpSched gs_single, gs_local;              // one-GPU schedule; multi-GPU: the rank's own subtrees
    std::vector<GrpSched> gs_stage;            // multi-GPU: the replicated fronts this rank holds, per exchange step (sn_gdepth)
    GrpSched* gs_cur = nullptr;                // ... the step launch_fronts is working on
    std::vector<hipEvent_t> la_evA, la_evB;
    hipStream_t stream3 = nullptr;             // third stream: the side buckets of small fronts next to a level's main launches (enqueue_factor)
    hipStream_t stream2 = nullptr; bool la_pending = false; hipEvent_t la_last = nullptr; bool lookahead = true, la_any = false; int la_wgs = 1 << 20, la_min_nt = 8;
    std::vector<size_t> reg_lds;
    std::vector<int> mid_split; std::vector<size_t> mid_lds;   // per level: leading FC_LDS128 fronts of order <= 96 (6x6-tile kernel, 2 workgroups per CU) and their LDS need
    std::vector<int> tiny_split;      // per level: number of leading FC_WAVE fronts of order <= 16 that use the 2x2-tile kernel
    std::vector<int> tiny16;          // per level: leading FC_WAVE fronts of order <= 16 (the four-per-wavefront static-order kernel takes them first)
    // multi-GPU schedules: buckets (level, class) of the fronts this rank owns / of the replicated top, stored behind
    // the single-GPU list in the same device array
    struct Sched { std::vector<int> ptr; int base = 0; std::vector<int> maxm, maxk, tiles, tiles64, last0, last1; std::vector<char> allsolo; };   // last0/1: per level, the group-last BIG fronts (solve units)
    Sched sch_local; std::vector<Sched> sch_stage;      // sch_stage[d]: replicated fronts of exchange step d (ranges of ranks d bisections below the whole machine) that this rank holds
    // Exchange steps (subtree-to-subcube mapping; the classic replicated top is the case of ONE step): a replicated front is held by a range of
    // ranks; what its children OUTSIDE that range -- subtrees owned by one rank, fronts of a sub-range -- contribute travels through the front's
    // arena square / top-rhs accumulator, written by ONE reporting rank per child (its owner; the first rank of its range) and summed over the
    // ranks, all ranges of one depth in one collective, deepest first.  join[c]: the fronts this rank reports a child of kind c to
    // (c = 0: own subtree roots, c = 1 + d: fronts of its depth-d range) and the code those children carry in ChildMeta::owner.
    struct JoinList { int base = 0, count = 0, maxm = 0, who = -1; };
    std::vector<JoinList> join;
    int ndepth = 1;
    std::vector<long long> abeg, aend, tbeg, tend;            // per step: its part of the arena / of the top right-hand sides
    struct RangeSeg { int d, glo, gsz; long long abeg, aend, tbeg, tend; };      // ... and inside a step the part of every range of ranks [glo, glo + gsz)
    std::vector<RangeSeg> rsegs;
    long long arena_doubles = 0, toprhs_doubles = 0;
    bool multi = false;

    // ---- device-side value assembly: per segment a device source buffer + a pinned staging buffer of the same length ----
    AsmSegs asm_{};
    std::vector<double*> asm_host;        // pinned staging per segment
    double* asm_pool = nullptr; double* asm_hpool = nullptr;
    bool assembly_define(int nseg, const int64_t* off, const int64_t* len) {
        DeviceGuard guard(dev);
        if (!ready) { err_ = "assembly_define: analyse first (and a usable HIP device)"; return false; }
        if (nseg < 1 || nseg > ASM_MAXSEG) { err_ = "assembly_define: 1..16 segments"; return false; }
        long long total = 0;
        for (int q = 0; q < nseg; ++q) { if (off[q] != total || len[q] < 0) { err_ = "assembly_define: segments must tile [0, nnz) in order"; return false; } total += len[q]; }
        if (total != S->nnz_in) { err_ = "assembly_define: segments do not cover the nnz triplet values"; return false; }
        if (asm_pool) { (void)hipFree(asm_pool); asm_pool = nullptr; }
        if (asm_hpool) { (void)hipHostFree(asm_hpool); asm_hpool = nullptr; }
        HIPCHK(hipMalloc((void**)&asm_pool, std::max<long long>(total, 1) * sizeof(double)));
        HIPCHK(hipMemset(asm_pool, 0, std::max<long long>(total, 1) * sizeof(double)));
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipHostMalloc((void**)&asm_hpool, std::max<long long>(total, 1) * sizeof(double), hipHostMallocDefault));
        asm_.nseg = nseg; asm_host.assign(nseg, nullptr);
        for (int q = 0; q < nseg; ++q) { asm_.off[q] = off[q]; asm_.len[q] = len[q]; asm_.src[q] = asm_pool + off[q]; asm_host[q] = asm_hpool + off[q]; asm_.scale[q] = 0.0; asm_.shift[q] = 0.0; }
        return true;
    }
    double* assembly_buffer(int seg) { return (seg >= 0 && seg < asm_.nseg) ? asm_host[seg] : nullptr; }
    bool assembly_upload(int seg) {
        DeviceGuard guard(dev);
        if (seg < 0 || seg >= asm_.nseg) { err_ = "assembly_upload: no such segment"; return false; }
        if (asm_.len[seg] > 0) HIPCHK(hipMemcpyAsync((void*)asm_.src[seg], asm_host[seg], (size_t)asm_.len[seg] * sizeof(double), hipMemcpyHostToDevice, stream));
        asm_dirty = true;
        return true;
    }
    bool factor_assembled(const double* scale, const double* shift, FactorStats& st) {
        {
            DeviceGuard guard(dev);
            if (!ready || asm_.nseg == 0) { err_ = "factor_assembled: assembly_define first"; return false; }
            long long mx = 1;
            bool same_scale = !asm_dirty;
            for (int q = 0; q < asm_.nseg; ++q) { same_scale = same_scale && asm_prev_scale[q] == scale[q]; asm_prev_scale[q] = scale[q]; }
            keep_scale_now = same_scale && opt.scaling == 1 && scale_valid && !knob_disabled("keep_scale");
            asm_dirty = false;
            for (int q = 0; q < asm_.nseg; ++q) { asm_.scale[q] = scale[q]; asm_.shift[q] = shift[q]; mx = std::max(mx, asm_.len[q]); }
            hipLaunchKernelGGL(k_assemble_segments, dim3(grid1d(mx), asm_.nseg), dim3(256), 0, stream, (double*)V.tvals, asm_);
            HIPCHK(hipGetLastError());
            have_values = true;
        }
        const bool ok = factor(nullptr, true, st);       // the values are on the device: the "refactor" path, no host buffer involved
        keep_scale_now = false;
        return ok;
    }

    // ---- communicator of a multi-GPU handle: RCCL over xGMI created from an ncclUniqueId, or a caller-supplied
    //      all-reduce (a host with its own communication layer; the single-GPU multi-rank tests).  All collectives are
    //      enqueued on the solver's stream: no host synchronisation between factor_local -> all-reduce -> factor_top. ----
    struct Rccl {
        void* lib = nullptr; ncclComm_t comm = nullptr;
        ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
        ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
        ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
        ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
        ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t*, void*) = nullptr;      // (RCCL >= 2.18; optional)
        ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;                         // (optional: what the communicator itself says its size is)
        const char* (*GetErrorString)(ncclResult_t) = nullptr;
        std::vector<ncclComm_t> sub;       // per exchange step: the communicator of the range of ranks this rank belongs to there (null: none / the whole machine)
        bool sub_ok = false;
    } rccl;
    int (*comm_range_fn)(void*, void*, int64_t, int, void*, int, int) = nullptr;
    int comm_kind = 0;                     // 0 none, 1 callback, 2 RCCL
    int (*comm_fn)(void*, void*, int64_t, int, void*) = nullptr; void* comm_ctx = nullptr;
    static bool rccl_load(Rccl& R, std::string& err) {
        if (R.lib) return true;
        // an RCCL instance the host process already loaded (e.g. the one bundled with PyTorch, soname librccl.so.1) is reused:
        // one library, several communicators -- never two RCCL copies in one process
        R.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!R.lib) R.lib = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
        if (!R.lib) R.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!R.lib) R.lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!R.lib) { err = std::string("dlopen(librccl.so): ") + dlerror(); return false; }
        R.GetUniqueId = (decltype(R.GetUniqueId))dlsym(R.lib, "ncclGetUniqueId");
        R.CommInitRank = (decltype(R.CommInitRank))dlsym(R.lib, "ncclCommInitRank");
        R.AllReduce = (decltype(R.AllReduce))dlsym(R.lib, "ncclAllReduce");
        R.CommDestroy = (decltype(R.CommDestroy))dlsym(R.lib, "ncclCommDestroy");
        R.CommSplit = (decltype(R.CommSplit))dlsym(R.lib, "ncclCommSplit");
        R.CommCount = (decltype(R.CommCount))dlsym(R.lib, "ncclCommCount");
        R.GetErrorString = (decltype(R.GetErrorString))dlsym(R.lib, "ncclGetErrorString");
        if (!R.GetUniqueId || !R.CommInitRank || !R.AllReduce || !R.CommDestroy) { err = "librccl.so lacks the nccl* entry points"; return false; }
        return true;
    }
    bool set_comm_rccl(const void* id128) {
        DeviceGuard guard(dev);
        if (!ready || !multi) { err_ = "set_comm_rccl: not a multi-GPU handle (nranks > 1 at create, analyse first)"; return false; }
        if (!rccl_load(rccl, err_)) return false;
        destroy_subcomms();
        if (rccl.comm) { (void)rccl.CommDestroy(rccl.comm); rccl.comm = nullptr; }
        ncclUniqueId id; std::memcpy(&id, id128, sizeof(id));
        ncclResult_t r = rccl.CommInitRank(&rccl.comm, opt.nranks, id, opt.rank);
        if (r != ncclSuccess) { err_ = std::string("ncclCommInitRank: ") + (rccl.GetErrorString ? rccl.GetErrorString(r) : "error"); rccl.comm = nullptr; return false; }
        comm_kind = 2;
        make_subcomms();
        return true;
    }
    // One sub-communicator per exchange step for the range of ranks this rank belongs to there (ncclCommSplit: collective over the whole
    // communicator, every rank calls it once per step that has a range smaller than the machine -- the symbolic structures are identical on
    // every rank, so they agree on which steps those are).  Without ncclCommSplit (or if a split fails) the steps fall back to ONE all-reduce
    // over the whole communicator, ranks outside a range contributing zeros: correct, only more traffic.
    void destroy_subcomms() {
        for (ncclComm_t c : rccl.sub) if (c && rccl.CommDestroy) (void)rccl.CommDestroy(c);
        rccl.sub.clear(); rccl.sub_ok = false;
    }
    void make_subcomms() {
        destroy_subcomms();
        if (comm_kind != 2 || !rccl.comm || !rccl.CommSplit || knob_disabled("subcomm")) return;
        rccl.sub.assign(ndepth, nullptr);
        bool ok = true;
        for (int d = 0; d < ndepth; ++d) {
            bool partial = false; int color = -1;      // NCCL_SPLIT_NOCOLOR
            for (const RangeSeg& sg : rsegs) if (sg.d == d && sg.gsz < opt.nranks) { partial = true; if (sg.glo <= opt.rank && opt.rank < sg.glo + sg.gsz) color = sg.glo; }
            if (!partial) continue;
            ncclComm_t nc = nullptr;
            if (rccl.CommSplit(rccl.comm, color, opt.rank, &nc, nullptr) != ncclSuccess) ok = false;
            rccl.sub[d] = nc;
        }
        // (a failed split on one rank may have succeeded on another: the fallback must be taken by everybody, which a sum over the ranks decides)
        int* flag = d_stats;      // scratch int on the device
        const int bad = ok ? 0 : 1;
        if (hipMemcpyAsync(flag, &bad, sizeof(int), hipMemcpyHostToDevice, stream) == hipSuccess &&
            rccl.AllReduce(flag, flag, 1, ncclInt32, ncclSum, rccl.comm, stream) == ncclSuccess) {
            int tot = 1;
            if (hipMemcpyAsync(&tot, flag, sizeof(int), hipMemcpyDeviceToHost, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess) rccl.sub_ok = tot == 0;
        }
        if (!rccl.sub_ok) { for (ncclComm_t& c : rccl.sub) { if (c && rccl.CommDestroy) (void)rccl.CommDestroy(c); c = nullptr; } }
        if (opt.verbose) fprintf(stderr, "[mi355x_kkt] rank %d: range-local collectives %s\n", opt.rank, rccl.sub_ok ? "on (ncclCommSplit)" : "off (whole-communicator all-reduce per step)");
    }
    bool set_comm_range_callback(int (*fn)(void*, void*, int64_t, int, void*, int, int)) { comm_range_fn = fn; return true; }
    bool range_local() const { return (comm_kind == 2 && rccl.sub_ok) || (comm_kind == 1 && comm_range_fn != nullptr); }
    // in-place sum of `count` elements over the ranks [glo, glo + gsz) (this rank is one of them), stream-ordered
    bool allreduce_range(void* dptr, long long count, int dtype, int d, int glo, int gsz) {
        if (count <= 0) return true;
        if (gsz >= opt.nranks) return allreduce(dptr, count, dtype);
        if (comm_kind == 2) {
            ncclComm_t c = rccl.sub[d];
            if (!c) { err_ = "internal: no sub-communicator for a range this rank belongs to"; return false; }
            ncclResult_t r = rccl.AllReduce(dptr, dptr, (size_t)count, dtype == 0 ? ncclDouble : ncclInt32, ncclSum, c, stream);
            if (r != ncclSuccess) { err_ = std::string("ncclAllReduce (range): ") + (rccl.GetErrorString ? rccl.GetErrorString(r) : "error"); return false; }
            return true;
        }
        if (comm_range_fn(comm_ctx, dptr, (int64_t)count, dtype, (void*)stream, glo, gsz) != 0) { err_ = "range all-reduce callback failed"; return false; }
        return true;
    }
    // the exchange of one step: its arena squares (what == 0) or top right-hand sides (what == 1) summed over the ranks that hold the fronts
    bool exchange_step(int d, int what) {
        double* base = what == 0 ? V.arena : V.top_rhs;
        if (range_local()) {
            for (const RangeSeg& sg : rsegs) {
                if (sg.d != d || !(sg.glo <= opt.rank && opt.rank < sg.glo + sg.gsz)) continue;
                const long long b = what == 0 ? sg.abeg : sg.tbeg, e = what == 0 ? sg.aend : sg.tend;
                if (!allreduce_range(base + b, e - b, 0, d, sg.glo, sg.gsz)) return false;
            }
            return true;
        }
        const long long b = what == 0 ? abeg[d] : tbeg[d], e = what == 0 ? aend[d] : tend[d];
        return e > b ? allreduce(base + b, e - b, 0) : true;
    }
    bool set_comm_callback(int (*fn)(void*, void*, int64_t, int, void*), void* ctx) {
        if (!ready || !multi) { err_ = "set_comm_callbacks: not a multi-GPU handle (nranks > 1 at create, analyse first)"; return false; }
        comm_fn = fn; comm_ctx = ctx; comm_kind = fn ? 1 : 0; return true;
    }
    // in-place sum over the ranks of `count` elements of device memory (dtype 0: fp64, 1: int32), stream-ordered
    bool allreduce(void* dptr, long long count, int dtype) {
        if (count <= 0) return true;
        if (comm_kind == 2) {
            ncclResult_t r = rccl.AllReduce(dptr, dptr, (size_t)count, dtype == 0 ? ncclDouble : ncclInt32, ncclSum, rccl.comm, stream);
            if (r != ncclSuccess) { err_ = std::string("ncclAllReduce: ") + (rccl.GetErrorString ? rccl.GetErrorString(r) : "error"); return false; }
            return true;
        }
        if (comm_kind == 1) {
            if (comm_fn(comm_ctx, dptr, (int64_t)count, dtype, (void*)stream) != 0) { err_ = "all-reduce callback failed"; return false; }
            return true;
        }
        err_ = "multi-GPU handle without a communicator: call mi355x_kkt_set_comm_rccl / _set_comm_callbacks (or drive the phase entry points yourself)";
        return false;
    }

    // ---- per-kernel-kind timing (bench.py roofline): hip events around every launch, eager mode ----
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev; std::vector<int> prof_kind; size_t prof_used = 0;
    double prof_ms[KK_COUNT] = {0}; int prof_launches[KK_COUNT] = {0};
    void prof_begin(int kind, hipStream_t strm = nullptr) {
        if (!prof_on) return;
        if (prof_used + 2 > prof_ev.size()) { size_t old = prof_ev.size(); prof_ev.resize(old + 64); for (size_t i = old; i < prof_ev.size(); ++i) (void)hipEventCreate(&prof_ev[i]); }
        (void)hipEventRecord(prof_ev[prof_used], strm ? strm : stream); prof_kind.push_back(kind);
    }
    void prof_end(hipStream_t strm = nullptr) { if (!prof_on) return; (void)hipEventRecord(prof_ev[prof_used + 1], strm ? strm : stream); prof_used += 2; }
    void prof_collect() {
        (void)hipStreamSynchronize(stream);
        for (size_t i = 0; i < prof_used; i += 2) { float ms = 0; (void)hipEventElapsedTime(&ms, prof_ev[i], prof_ev[i + 1]); int kd = prof_kind[i / 2]; prof_ms[kd] += ms; prof_launches[kd]++; }
        prof_used = 0; prof_kind.clear();
    }
    ~NumericImpl() { release(); for (auto e : prof_ev) (void)hipEventDestroy(e); }
    // keep = true (restructure): what does not depend on the elimination structure survives -- the communicator, the streams, the pinned
    // staging buffers handed out to the caller (values_buffer, assembly buffers), the device copy of the triplet values, the assembly
    // sources, the primal-dual workspace (it reads the triplet values and calls solve), the caller's scaling factors (original numbering)
    double* keep_tvals = nullptr;
    double* kept_pool = nullptr; long long kept_pool_doubles = 0;      // the one-piece L | cb pool: survives a restructure (a delayed-pivot edit changes the layout by a fraction of a per cent: the 1 % of slack it is allocated with takes that), 40-80 ms of hipFree + hipMalloc at 9 GiB less per edit
    void release(bool keep = false) {
        DeviceGuard guard(dev);
        destroy_subcomms();                      // (the ranges follow the structure: split again after a restructure)
        sctx_release();
        if (!keep) { if (rccl.comm && rccl.CommDestroy) { (void)rccl.CommDestroy(rccl.comm); rccl.comm = nullptr; }
                     comm_kind = 0; comm_range_fn = nullptr; }
        destroy_factor_graphs(); scale_valid = false; asm_dirty = true;
        if (g_solve) { (void)hipGraphExecDestroy(g_solve); g_solve = nullptr; }
        keep_tvals = nullptr;
        for (void* p : allocs) {
            if (keep && p == (void*)V.tvals) { keep_tvals = (double*)p; continue; }
            if (keep && kept_pool && p == (void*)kept_pool) continue;      // (one-piece pool: handed to the next set-up, which reuses it if the new layout fits)
            if (keep && p == (void*)d_user_scale) continue;
            (void)hipFree(p);
        }
        allocs.clear();
        if (!keep && kept_pool) { kept_pool = nullptr; kept_pool_doubles = 0; }      // (it was in allocs: freed above)
        if (keep && keep_tvals) allocs.push_back(keep_tvals);
        if (keep && d_user_scale) allocs.push_back(d_user_scale); else d_user_scale = nullptr;
        if (d_rhs) { (void)hipFree(d_rhs); d_rhs = nullptr; d_rhs_cap = 0; }
        match_free();
        if (!keep) {
            if (asm_pool) { (void)hipFree(asm_pool); asm_pool = nullptr; }
            if (asm_hpool) { (void)hipHostFree(asm_hpool); asm_hpool = nullptr; }
            asm_.nseg = 0;
            pd_free();
            if (h_vals) { (void)hipHostFree(h_vals); h_vals = nullptr; }
            if (h_stats) { (void)hipHostFree(h_stats); h_stats = nullptr; }
            if (ev0) { (void)hipEventDestroy(ev0); ev0 = nullptr; }
            if (ev1) { (void)hipEventDestroy(ev1); ev1 = nullptr; }
        }
        for (auto e : la_evA) if (e) (void)hipEventDestroy(e);
        for (auto e : la_evB) if (e) (void)hipEventDestroy(e);
        la_evA.clear(); la_evB.clear();
        { std::vector<GrpSched*> all{&gs_single, &gs_local}; for (auto& g : gs_stage) all.push_back(&g);
          for (GrpSched* g : all) { for (auto* v : {&g->evA, &g->evB}) { for (auto e : *v) if (e) (void)hipEventDestroy(e); v->clear(); } } }
        for (auto* v : {&side_evF, &side_evJ}) { for (auto e : *v) if (e) (void)hipEventDestroy(e); v->clear(); }
        la_last = nullptr; la_pending = false;
        if (!keep) {
            if (stream3) { (void)hipStreamDestroy(stream3); stream3 = nullptr; }
            if (stream2) { (void)hipStreamDestroy(stream2); stream2 = nullptr; }
            if (stream) { (void)hipStreamDestroy(stream); stream = nullptr; }
        }
        arena_doubles = 0; toprhs_doubles = 0;
        ready = false;
    }
    template <class T, class A> bool upload(const std::vector<T, A>& h, const T** d) {
        T* p = nullptr; size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(T);
        HIPCHK(hipMalloc((void**)&p, bytes)); allocs.push_back(p);
        if (!h.empty()) HIPCHK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
        *d = p; return true;
    }
    template <class T> bool dalloc(T** d, size_t count) {
        T* p = nullptr; HIPCHK(hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T))); allocs.push_back(p);
        HIPCHK(hipMemset(p, 0, std::max<size_t>(count, 1) * sizeof(T)));
        *d = p; return true;
    }

    // The layout of the exchange steps (a function of the symbolic structure only, the same on every rank; no device needed -- the C ABI's
    // mi355x_kkt_comm_plan walks it on a machine without a GPU): top-rhs accumulators for every replicated front; arena squares only for those
    // with a child from outside their range (the joins): that is all the all-reduce has to carry (A is replicated input, not reduced).  Only the
    // LOWER triangle of a square travels (packed by columns, the layout the front kernels assemble into), and inside a step the squares are
    // grouped by the range of ranks that holds their front: what a front receives comes from ranks of its own range only, so a range sums its
    // part among its own ranks (sub-communicator / range callback) -- the other ranks neither send nor receive it.
    static void exchange_layout(const Symbolic& Sy, int ndepth, std::vector<long long>& aoff, std::vector<long long>& troff, std::vector<long long>& abeg, std::vector<long long>& aend,
                                std::vector<long long>& tbeg, std::vector<long long>& tend, std::vector<RangeSeg>& rsegs, long long& arena_doubles, long long& toprhs_doubles) {
        auto same_range = [&](int a, int b) { return Sy.sn_owner[a] < 0 && Sy.sn_owner[b] < 0 && Sy.sn_glo[a] == Sy.sn_glo[b] && Sy.sn_gsz[a] == Sy.sn_gsz[b]; };
        auto crosses = [&](int c) { const int pa = Sy.sn_parent[c]; return pa >= 0 && Sy.sn_owner[pa] < 0 && !same_range(c, pa); };
        aoff.assign(Sy.num_sn, -1); troff.assign(Sy.num_sn, -1);
        arena_doubles = 0; toprhs_doubles = 0;
        std::vector<char> is_join(Sy.num_sn, 0);
        for (int c = 0; c < Sy.num_sn; ++c) if (crosses(c)) is_join[Sy.sn_parent[c]] = 1;
        abeg.assign(ndepth, 0); aend.assign(ndepth, 0); tbeg.assign(ndepth, 0); tend.assign(ndepth, 0);
        rsegs.clear();
        for (int d = 0; d < ndepth; ++d) {
            abeg[d] = arena_doubles; tbeg[d] = toprhs_doubles;
            std::vector<std::pair<int, int>> ranges;
            for (int s = 0; s < Sy.num_sn; ++s) if (Sy.sn_owner[s] < 0 && Sy.sn_gdepth[s] == d) {
                const std::pair<int, int> rg(Sy.sn_glo[s], Sy.sn_gsz[s]);
                if (std::find(ranges.begin(), ranges.end(), rg) == ranges.end()) ranges.push_back(rg);
            }
            std::sort(ranges.begin(), ranges.end());
            for (const auto& rg : ranges) {
                RangeSeg sg{d, rg.first, rg.second, arena_doubles, 0, toprhs_doubles, 0};
                for (int s = 0; s < Sy.num_sn; ++s) if (Sy.sn_owner[s] < 0 && Sy.sn_gdepth[s] == d && Sy.sn_glo[s] == rg.first && Sy.sn_gsz[s] == rg.second) {
                    const long long m = Sy.sn_rowptr[s + 1] - Sy.sn_rowptr[s];
                    troff[s] = toprhs_doubles; toprhs_doubles += m;
                    if (is_join[s]) { aoff[s] = arena_doubles; arena_doubles += m * (m + 1) / 2; }
                }
                sg.aend = arena_doubles; sg.tend = toprhs_doubles;
                rsegs.push_back(sg);
            }
            aend[d] = arena_doubles; tend[d] = toprhs_doubles;
        }
    }
    // The collectives ONE rank issues, in order, for a factorisation and for one solve -- and the ncclCommSplit calls of make_subcomms() before them:
    // records of 6 ints {what, depth, colour, range size, count, dtype}; what: 0 = ncclCommSplit (colour -1: NCCL_SPLIT_NOCOLOR, key = rank), 1 = all-reduce of
    // arena squares, 2 = inertia / pivot statistics, 3 = all-reduce of top right-hand sides, 4 = solution pieces; colour = first rank of the range (sub-communicator
    // of that step) or -2 = the whole communicator.  The SAME walk as factor_dist / solve_dist / make_subcomms / exchange_step, without a device.
    static void comm_plan(const Symbolic& Sy, int nranks, int rank, bool range_local, int n, std::vector<int>& out) {
        const int ndepth = std::max(1, Sy.num_gdepths);
        std::vector<long long> aoff, troff, abeg, aend, tbeg, tend; std::vector<RangeSeg> rsegs; long long ad = 0, td = 0;
        exchange_layout(Sy, ndepth, aoff, troff, abeg, aend, tbeg, tend, rsegs, ad, td);
        auto rec = [&](int what, int d, int colour, int gsz, long long count, int dtype) { out.push_back(what); out.push_back(d); out.push_back(colour); out.push_back(gsz); out.push_back((int)std::min<long long>(count, 0x7fffffffll)); out.push_back(dtype); };
        if (range_local)
            for (int d = 0; d < ndepth; ++d) {
                bool partial = false; int color = -1;
                for (const RangeSeg& sg : rsegs) if (sg.d == d && sg.gsz < nranks) { partial = true; if (sg.glo <= rank && rank < sg.glo + sg.gsz) color = sg.glo; }
                if (partial) rec(0, d, color, 0, 0, 1);
            }
        auto step = [&](int d, int what) {
            if (range_local) {
                for (const RangeSeg& sg : rsegs) {
                    if (sg.d != d || !(sg.glo <= rank && rank < sg.glo + sg.gsz)) continue;
                    const long long cnt = what == 1 ? sg.aend - sg.abeg : sg.tend - sg.tbeg;
                    if (cnt > 0) rec(what, d, sg.gsz >= nranks ? -2 : sg.glo, sg.gsz, cnt, 0);
                }
            } else {
                const long long cnt = what == 1 ? aend[d] - abeg[d] : tend[d] - tbeg[d];
                if (cnt > 0) rec(what, d, -2, nranks, cnt, 0);
            }
        };
        for (int d = ndepth - 1; d >= 0; --d) step(d, 1);
        rec(2, 0, -2, nranks, 8, 1);
        for (int d = ndepth - 1; d >= 0; --d) step(d, 3);
        if (n > 0) rec(4, 0, -2, nranks, n, 0);
    }
    // Delayed pivots changed the structure (symbolic.cpp restructure_delays): everything derived from it is rebuilt, the rest (see release) stays
    bool restructure(const Symbolic& Sy) {
        if (!have_device || !stream) { err_ = "restructure: solver not set up"; return false; }
        { DeviceGuard guard(dev); (void)hipStreamSynchronize(stream); if (stream2) (void)hipStreamSynchronize(stream2); if (stream3) (void)hipStreamSynchronize(stream3); }
        const NumericOptions o = opt;
        const bool hv = have_values;
        if (!setup(Sy, o, true)) return false;
        have_values = hv;
        return true;
    }
    bool setup(const Symbolic& Sy, const NumericOptions& o, bool keep = false) {
        auto now_ = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double t_prev = now_();
        release(keep); S = &Sy; opt = o;
        if (o.verbose >= 2 && keep) fprintf(stderr, "[mi355x_kkt]   setup %-28s %.3f s\n", "release of the old structure", now_() - t_prev);
        t_prev = now_();
        auto lap = [&](const char* what) { if (opt.verbose >= 2) { const double t = now_(); fprintf(stderr, "[mi355x_kkt]   setup %-28s %.3f s\n", what, t - t_prev); t_prev = t; } };
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
            err_ = "no HIP device available: the MI355X KKT backend has no CPU fallback"; have_device = false; return false; }
        if (keep) { /* the device of the first setup */ } else if (opt.device >= 0) dev = opt.device; else HIPCHK(hipGetDevice(&dev));
        DeviceGuard guard(dev);
        have_device = true;
        if (!keep) {   // the main stream carries the latency-bound pivot chains: highest priority; the look-ahead stream the lowest
            int plo = 0, phi = 0;
            (void)hipDeviceGetStreamPriorityRange(&plo, &phi);
            HIPCHK(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, phi));
            HIPCHK(hipStreamCreateWithPriority(&stream2, hipStreamNonBlocking, plo));
            HIPCHK(hipStreamCreateWithPriority(&stream3, hipStreamNonBlocking, (plo + phi) / 2));
        }
        lookahead = !knob_disabled("lookahead");
        la_wgs = (int)std::max(1ll, knob_int("la_wgs", la_wgs));          // development knobs
        la_min_nt = (int)std::max(3ll, knob_int("la_min_nt", la_min_nt));
        if (!keep) {
            HIPCHK(hipEventCreate(&ev0)); HIPCHK(hipEventCreate(&ev1));
            if (opt.prewarmed_vals && opt.prewarmed_count >= std::max<size_t>(Sy.nnz_in, 1)) h_vals = (double*)opt.prewarmed_vals;       // (made while the analysis ran)
            else {
                if (opt.prewarmed_vals) (void)hipHostFree(opt.prewarmed_vals);
                HIPCHK(hipHostMalloc((void**)&h_vals, std::max<size_t>(Sy.nnz_in, 1) * sizeof(double), hipHostMallocDefault));
            }
            HIPCHK(hipHostMalloc((void**)&h_stats, 12 * sizeof(int), hipHostMallocDefault));
        }
        opt.prewarmed_vals = nullptr;
        lap("device, streams, pinned buffer");
        {   // Everything numeric is allocated once, here; the bulk is ONE pool  L | cb  (every contribution block resident: DESIGN.md "Data layout").  A structure
            // whose pool does not fit -- a 3-D problem beyond MBndryCntrl_3D N ~ 120 on 288 GB, or a device shared with another process -- is refused with
            // the numbers, before the allocation is attempted: factor() / solve() then answer MI355X_KKT_FATAL with this message (no partial set-up, no
            // fallback).  MI355X_KKT_POOL_LIMIT_GIB caps what one handle may take (a GPU shared by several ranks or applications).
            size_t fr = 0, tot = 0;
            const bool have_meminfo = hipMemGetInfo(&fr, &tot) == hipSuccess;      // (a failed query must not read as "0 bytes free": the fit check is then left to the allocation itself)
            if (!have_meminfo) { (void)hipGetLastError(); fr = ~(size_t)0 >> 1; }
            const double gib = 1.0 / (1024.0 * 1024.0 * 1024.0);
            const double need = 8.0 * ((double)Sy.l_doubles + (double)Sy.cb_doubles + (double)Sy.wbuf_doubles + (double)Sy.minv_doubles + (double)Sy.cvec_doubles + (double)Sy.gpart_doubles) +
                                12.0 * (double)Sy.nnz_a + 16.0 * (double)Sy.nnz_in + 12.0 * (double)Sy.rslot_idx.size() + 200.0 * (double)Sy.n;
            double cap = (double)fr + (kept_pool ? 8.0 * (double)kept_pool_doubles : 0.0);      // (a restructure: the pool of the structure before the edit is still held, and will be reused or freed)
            if (const char* e = getenv("MI355X_KKT_POOL_LIMIT_GIB")) cap = std::min(cap, atof(e) / gib);
            if (need > cap) {
                char msg[512];
                snprintf(msg, sizeof msg, "the factor + contribution-block pool of this structure does not fit the device: %.2f GiB needed (L %.2f + contribution blocks %.2f + work space), "
                                          "%.2f GiB available (%.2f GiB free of %.2f%s); no out-of-core path, no CPU fallback", need * gib, 8.0 * (double)Sy.l_doubles * gib, 8.0 * (double)Sy.cb_doubles * gib,
                         cap * gib, (double)fr * gib, (double)tot * gib, getenv("MI355X_KKT_POOL_LIMIT_GIB") ? ", capped by MI355X_KKT_POOL_LIMIT_GIB" : "");
                err_ = msg; return false;
            }
        }
        // The pool in PIECES (round 5; MI355X_KKT_POOL_PIECE_MIB, default: one block).  hipMalloc of ONE block takes 0.5 s at 16 GiB, 1.4 s at 32, 2.2-2.7 s at the 74 GiB
        // of MBndryCntrl_3D 78 (tools/micro/malloc_time.hip) -- the largest item of that run's whole set-up -- and ten blocks of 7.4 GiB took 3 ms in the micro benchmark,
        // but only because the process had held and freed that memory before: in a fresh process twelve pieces cost what the one block costs (1.6-2.1 s: the price is per
        // byte mapped, whoever asks; a virtual range backed by hipMemCreate / hipMemMap pieces: 4.3 s).  Kept as an option for a device too fragmented for one block.
        // The layout  L | cb  stays ONE linear offset space (symbolic.cpp step 12); it is cut between the blocks of two fronts (the panel or the contribution block of
        // a front that is not in place on a child: everything an in-place chain touches lies inside its first link's block) into segments of at most `piece` doubles,
        // every segment is its own allocation, and an offset of the symbolic structure becomes  offset + (address of its segment - address of segment 0) / 8 - (start
        // of the segment):  V.L + offset then points into the right allocation; every table the kernels read is built from PO() / CO() below.
        std::vector<long long> pool_cut, pool_delta;
        {
            const long long total = Sy.l_doubles + Sy.cb_doubles;
            long long piece = 1ll << 60;      // default: ONE block (see above: pieces do not make the allocation faster)
            if (const char* e = getenv("MI355X_KKT_POOL_PIECE_MIB")) piece = std::max(1ll, atoll(e)) << 17;      // (MiB -> doubles; a device too fragmented for one block, tests)
            std::vector<long long> starts;
            starts.reserve(2 * (size_t)Sy.num_sn + 1);
            for (int sn = 0; sn < Sy.num_sn; ++sn) if (Sy.alias_child[sn] < 0) { starts.push_back(Sy.panel_off[sn]); starts.push_back(Sy.l_doubles + Sy.cb_off[sn]); }
            starts.push_back(total);
            std::sort(starts.begin(), starts.end());
            long long cur = 0, last_ok = 0;
            pool_cut.push_back(0);
            for (long long b : starts) {
                if (b <= cur) continue;
                while (b - cur > piece && last_ok > cur) { cur = last_ok; pool_cut.push_back(cur); }      // (a single block larger than a piece gets a segment of its own size)
                last_ok = b;
            }
            std::vector<char*> base(pool_cut.size(), nullptr);
            if (kept_pool && (pool_cut.size() != 1 || total > kept_pool_doubles)) { (void)hipFree(kept_pool); kept_pool = nullptr; kept_pool_doubles = 0; }
            for (size_t i = 0; i < pool_cut.size(); ++i) {
                const long long end = i + 1 < pool_cut.size() ? pool_cut[i + 1] : total;
                double* pp = nullptr;
                if (pool_cut.size() == 1 && kept_pool) {      // the pool of the structure before the edit: reused, zeroed like a fresh one
                    pp = kept_pool; allocs.push_back(pp);
                    HIPCHK(hipMemsetAsync(pp, 0, (size_t)std::max<long long>(total, 1) * sizeof(double), stream));
                } else if (pool_cut.size() == 1) {
                    const long long cap = total + total / 100 + 1024;
                    if (!dalloc(&pp, (size_t)cap)) return false;
                    kept_pool = pp; kept_pool_doubles = cap;
                } else if (!dalloc(&pp, (size_t)std::max<long long>(end - pool_cut[i], 1))) return false;
                base[i] = (char*)pp;
                pool_delta.push_back((long long)((base[i] - base[0]) / (ptrdiff_t)sizeof(double)) - pool_cut[i]);
            }
            V.L = (double*)base[0];
            if (opt.verbose >= 2) fprintf(stderr, "[mi355x_kkt]   pool: %.2f GiB in %zu pieces\n", 8.0 * (double)total / (1024.0 * 1024.0 * 1024.0), pool_cut.size());
        }
        auto pool_remap = [&](long long lin) { const size_t i = (size_t)(std::upper_bound(pool_cut.begin(), pool_cut.end(), lin) - pool_cut.begin()) - 1; return lin + pool_delta[i]; };
        auto PO = [&](int sn) { return pool_remap(Sy.panel_off[sn]); };                                   // panel of a front, relative to V.L
        auto CO = [&](int sn) { return pool_remap(Sy.l_doubles + Sy.cb_off[sn]) - Sy.l_doubles; };        // contribution block, relative to V.cb = V.L + l_doubles
        lap("pool");
        std::vector<long long> poff(Sy.num_sn), coff(Sy.num_sn), moff(Sy.minv_off.begin(), Sy.minv_off.end());
        for (int sn = 0; sn < Sy.num_sn; ++sn) { poff[sn] = PO(sn); coff[sn] = CO(sn); }
        multi = opt.nranks > 1 || getenv("MI355X_KKT_FORCE_MULTI") != nullptr;   // (1-rank multi path: plumbing tests on a 1-GPU box)
        std::vector<int> lvl_list(Sy.level_sn.begin(), Sy.level_sn.end());
        std::vector<char> solve_entry;      // parallel to lvl_list: 1 = entry of a solve-unit list
        std::vector<long long> aoff(Sy.num_sn, -1), troff(Sy.num_sn, -1);
        std::vector<int> colown(Sy.n, 0);
        if (multi) {
            const int P = std::max(1, opt.nranks);
            ndepth = std::max(1, Sy.num_gdepths);
            auto held = [&](int s) { return Sy.sn_owner[s] < 0 && Sy.sn_glo[s] <= opt.rank && opt.rank < Sy.sn_glo[s] + Sy.sn_gsz[s]; };      // replicated front on this rank
            auto build = [&](Sched& sc, int depth) {      // depth < 0: the rank's own subtrees
                sc.ptr.assign((size_t)Sy.num_levels * FC_COUNT + 1, 0); sc.base = (int)lvl_list.size();
                sc.maxm.assign(Sy.num_levels, 0); sc.maxk.assign(Sy.num_levels, 0); sc.tiles.assign(Sy.num_levels, 0); sc.tiles64.assign(Sy.num_levels, 0);
                std::vector<std::vector<int>> bucket((size_t)Sy.num_levels * FC_COUNT);
                for (int s = 0; s < Sy.num_sn; ++s) {
                    const bool mine = depth >= 0 ? (held(s) && Sy.sn_gdepth[s] == depth) : (Sy.sn_owner[s] == opt.rank);
                    if (!mine) continue;
                    bucket[(size_t)Sy.sn_level[s] * FC_COUNT + Sy.sn_class[s]].push_back(s);
                    if (Sy.sn_class[s] == FC_BIG) { sc.maxm[Sy.sn_level[s]] = std::max(sc.maxm[Sy.sn_level[s]], Sy.sn_rowptr[s + 1] - Sy.sn_rowptr[s]);
                                                    sc.maxk[Sy.sn_level[s]] = std::max(sc.maxk[Sy.sn_level[s]], Sy.sn_colptr[s + 1] - Sy.sn_colptr[s]);
                                                    sc.tiles[Sy.sn_level[s]] = std::max(sc.tiles[Sy.sn_level[s]], schur_tiles(Sy, s));
                                                    sc.tiles64[Sy.sn_level[s]] = std::max(sc.tiles64[Sy.sn_level[s]], schur_tiles64(Sy, s)); }
                }
                for (size_t b = 0; b < bucket.size(); ++b) { sc.ptr[b + 1] = sc.ptr[b] + (int)bucket[b].size(); lvl_list.insert(lvl_list.end(), bucket[b].begin(), bucket[b].end()); }
                sc.last0.assign(Sy.num_levels, 0); sc.last1.assign(Sy.num_levels, 0);
                for (int lv = 0; lv < Sy.num_levels; ++lv) {
                    sc.last0[lv] = (int)lvl_list.size();
                    for (int sn : bucket[(size_t)lv * FC_COUNT + FC_BIG]) if (Sy.grp_rem[sn] == 0 || !Sy.solve_group) { lvl_list.push_back(sn); solve_entry.resize(lvl_list.size(), 0); solve_entry.back() = 1; }
                    sc.last1[lv] = (int)lvl_list.size();
                }
                sc.allsolo.assign(Sy.num_levels, 0);
                if (!Sy.solve_group)
                    for (int lv = 0; lv < Sy.num_levels; ++lv) {
                        bool all = sc.last1[lv] > sc.last0[lv];
                        for (int q = sc.last0[lv]; q < sc.last1[lv]; ++q) {
                            const int sn = lvl_list[q], nch = Sy.child_ptr[sn + 1] - Sy.child_ptr[sn];
                            if (!((Sy.alias_child[sn] >= 0 && nch == 1) || (Sy.alias_child[sn] < 0 && nch == 0))) all = false;
                        }
                        sc.allsolo[lv] = all ? 1 : 0;
                    }
            };
            build(sch_local, -1);
            sch_stage.assign(ndepth, Sched());
            for (int d = 0; d < ndepth; ++d) build(sch_stage[d], d);
            // a child goes through the arena / the top-rhs accumulators when its parent is a replicated front of ANOTHER range of ranks
            auto same_range = [&](int a, int b) { return Sy.sn_owner[a] < 0 && Sy.sn_owner[b] < 0 && Sy.sn_glo[a] == Sy.sn_glo[b] && Sy.sn_gsz[a] == Sy.sn_gsz[b]; };
            auto crosses = [&](int c) { const int pa = Sy.sn_parent[c]; return pa >= 0 && Sy.sn_owner[pa] < 0 && !same_range(c, pa); };
            exchange_layout(Sy, ndepth, aoff, troff, abeg, aend, tbeg, tend, rsegs, arena_doubles, toprhs_doubles);
            // what this rank reports: its own subtree roots (kind 0), and -- as the first rank of its depth-d range -- that range's fronts (kind 1 + d)
            join.assign(ndepth + 1, JoinList());
            for (int c = 0; c <= ndepth; ++c) {
                JoinList& J = join[c]; J.base = (int)lvl_list.size(); J.who = opt.rank + P * c;
                std::vector<char> listed(Sy.num_sn, 0);
                for (int ch = 0; ch < Sy.num_sn; ++ch) {
                    if (!crosses(ch)) continue;
                    const bool rep = c == 0 ? Sy.sn_owner[ch] == opt.rank : (Sy.sn_owner[ch] < 0 && Sy.sn_gdepth[ch] == c - 1 && Sy.sn_glo[ch] == opt.rank);
                    if (rep) listed[Sy.sn_parent[ch]] = 1;
                }
                for (int s = 0; s < Sy.num_sn; ++s) if (listed[s]) { lvl_list.push_back(s); ++J.count; J.maxm = std::max(J.maxm, Sy.sn_rowptr[s + 1] - Sy.sn_rowptr[s]); }
            }
            // the solution pieces are summed over the ranks: a column is reported by its owner / by the first rank of its front's range
            for (int s = 0; s < Sy.num_sn; ++s) {
                const int o = Sy.sn_owner[s] >= 0 ? Sy.sn_owner[s] : (Sy.sn_glo[s] == opt.rank ? opt.rank : P);
                for (int j = Sy.sn_colptr[s]; j < Sy.sn_colptr[s + 1]; ++j) colown[j] = o;
            }
        }
        // inside every (level, FC_WAVE) bucket of the single-GPU schedule: fronts of order <= 16 first.  When there are many of
        // them (throughput regime) they run on the 2x2-tile instantiation, whose small register footprint doubles the
        // number of resident wavefronts.
        tiny_split.assign(Sy.num_levels, 0); tiny16.assign(Sy.num_levels, 0);
        for (int lv = 0; lv < Sy.num_levels; ++lv) {
            const int b0 = Sy.level_ptr[(size_t)lv * FC_COUNT + FC_WAVE], b1 = Sy.level_ptr[(size_t)lv * FC_COUNT + FC_WAVE + 1];
            auto order_of = [&](int sn) { return Sy.sn_rowptr[sn + 1] - Sy.sn_rowptr[sn]; };
            std::stable_sort(lvl_list.begin() + b0, lvl_list.begin() + b1, [&](int a, int b) { return order_of(a) < order_of(b); });
            int q = b0; while (q < b1 && order_of(lvl_list[q]) <= 16) ++q;
            tiny_split[lv] = (q - b0 >= 2048) ? q - b0 : 0;
            tiny16[lv] = q - b0;
        }
        // same for the (level, FC_LDS128) buckets: fronts of order <= 96 first; they run on the 6x6-tile instantiation (half the
        // registers and LDS of the 8x8 one => two workgroups per CU)
        mid_split.assign(Sy.num_levels, 0); mid_lds.assign(Sy.num_levels, 0);
        for (int lv = 0; lv < Sy.num_levels; ++lv) {
            const int b0 = Sy.level_ptr[(size_t)lv * FC_COUNT + FC_LDS128], b1 = Sy.level_ptr[(size_t)lv * FC_COUNT + FC_LDS128 + 1];
            auto order_of = [&](int sn) { return Sy.sn_rowptr[sn + 1] - Sy.sn_rowptr[sn]; };
            std::stable_sort(lvl_list.begin() + b0, lvl_list.begin() + b1, [&](int a, int b) { return order_of(a) < order_of(b); });
            int q = b0;
            while (q < b1 && order_of(lvl_list[q]) <= 96) {
                const int sn = lvl_list[q];
                const size_t m = order_of(sn), k = Sy.sn_colptr[sn + 1] - Sy.sn_colptr[sn], ld = m | 1, ldi = k | 1;
                mid_lds[lv] = std::max(mid_lds[lv], (std::max(m * (m + 1) / 2, k * ld + k * ldi) + 4 * 96 + 3 * k) * sizeof(double) + 2 * k * sizeof(int) + 64);
                ++q;
            }
            mid_split[lv] = (q - b0 >= 256) ? q - b0 : 0;
        }
        // (level, FC_BIG) buckets of the single-GPU schedule: sorted by order and split at 1024 rows.  The two halves are
        // launched separately: tighter rectangular grids on heterogeneous levels, and the small fronts (a handful of tiles,
        // K = 16..64) take the 256-thread 64 x 64 trailing-update kernel while the large ones take the 1024-thread one.
        big_split.assign(Sy.num_levels, 0);
        for (int h = 0; h < 2; ++h) { part_mm[h].assign(Sy.num_levels, 0); part_kk[h].assign(Sy.num_levels, 0); part_tiles[h].assign(Sy.num_levels, 0); }
        for (int lv = 0; lv < Sy.num_levels; ++lv) {
            const int b0 = Sy.level_ptr[(size_t)lv * FC_COUNT + FC_BIG], b1 = Sy.level_ptr[(size_t)lv * FC_COUNT + FC_BIG + 1];
            auto order_of = [&](int sn) { return Sy.sn_rowptr[sn + 1] - Sy.sn_rowptr[sn]; };
            std::stable_sort(lvl_list.begin() + b0, lvl_list.begin() + b1, [&](int a, int b) { return order_of(a) < order_of(b); });
            int q = b0; while (q < b1 && order_of(lvl_list[q]) <= 1024) ++q;
            big_split[lv] = q - b0;
            for (int e = b0; e < b1; ++e) {
                const int sn = lvl_list[e], h = e < q ? 0 : 1;
                part_mm[h][lv] = std::max(part_mm[h][lv], order_of(sn));
                part_kk[h][lv] = std::max(part_kk[h][lv], Sy.sn_colptr[sn + 1] - Sy.sn_colptr[sn]);
                part_tiles[h][lv] = std::max(part_tiles[h][lv], h == 0 ? schur_tiles64(Sy, sn) : schur_tiles(Sy, sn));
            }
        }
        // single-GPU schedule: per level the group-last BIG fronts (the units of the triangular solves)
        big_last0.assign(Sy.num_levels, 0); big_last1.assign(Sy.num_levels, 0);
        for (int lv = 0; lv < Sy.num_levels; ++lv) {
            big_last0[lv] = (int)lvl_list.size();
            for (int q = Sy.level_ptr[(size_t)lv * FC_COUNT + FC_BIG]; q < Sy.level_ptr[(size_t)lv * FC_COUNT + FC_BIG + 1]; ++q)
                if (Sy.grp_rem[Sy.level_sn[q]] == 0 || !Sy.solve_group) { lvl_list.push_back(Sy.level_sn[q]); solve_entry.resize(lvl_list.size(), 0); solve_entry.back() = 1; }
            big_last1[lv] = (int)lvl_list.size();
        }
        lv_allsolo.assign(Sy.num_levels, 0);
        if (!Sy.solve_group)
            for (int lv = 0; lv < Sy.num_levels; ++lv) {
                bool all = big_last1[lv] > big_last0[lv];
                for (int q = big_last0[lv]; q < big_last1[lv]; ++q) {
                    const int sn = lvl_list[q], nch = Sy.child_ptr[sn + 1] - Sy.child_ptr[sn];
                    if (!((Sy.alias_child[sn] >= 0 && nch == 1) || (Sy.alias_child[sn] < 0 && nch == 0))) all = false;
                }
                lv_allsolo[lv] = all ? 1 : 0;
            }
        // ---- sync-free (data-flow) sweeps over the latency-bound top of the tree: a SEGMENT is a run of consecutive levels whose fronts are
        // all BIG with <= 64 pivots and of which there are at most chain_maxc per level; its fronts are cut into CHAINS (maximal runs of
        // in-place links: every link after the first has the previous link as its only child and shares its vector), and ONE launch per
        // sweep runs the whole segment: workgroups wait on flags for exactly what they consume (see k_fwd_chain / k_bwd_chain) ----
        std::vector<ChainLink> chl; std::vector<ChainDesc> chd; std::vector<int> chwait, wgf, wgb;
        int ntailflags = 0, ndots = 0;
        chain_segs.clear(); in_seg.clear(); seg_at_lv0.assign(Sy.num_levels, -1); seg_at_lv1.assign(Sy.num_levels, -1);
        chain_solve = !knob_disabled("chain_solve");
        fuse_dt = !knob_disabled("fuse_dt");
        V.fastpiv = !knob_disabled("fastpiv") ? 1 : 0;
        V.asm_pull = !knob_disabled("asm_pull") ? 1 : 0;
        grp_rbw_max = (int)std::max(1ll, knob_int("grp_rbw_max", grp_rbw_max));
        V.fastu = 1e-4; { double fl; if (knob_tune("fastpiv_floor", &fl)) V.fastu = fl; }      // (0.01 up to r03a: 9 % of the synth_1e6 blocks then took the strict loop and set the pace of their level: 23.1 -> 22.0 ms)
        fuse_dt_maxwg = (int)knob_int("fuse_dt_maxwg", fuse_dt_maxwg);
        pair_solve = !knob_disabled("pair_solve") && !multi;
        wave_kmax.assign(Sy.num_levels, 0); wave_mmax.assign(Sy.num_levels, 0); wave_mmin.assign(Sy.num_levels, 1 << 30);
        for (int sn = 0; sn < Sy.num_sn; ++sn) if (Sy.sn_class[sn] == FC_WAVE) {
            wave_kmax[Sy.sn_level[sn]] = std::max(wave_kmax[Sy.sn_level[sn]], Sy.sn_colptr[sn + 1] - Sy.sn_colptr[sn]);
            wave_mmax[Sy.sn_level[sn]] = std::max(wave_mmax[Sy.sn_level[sn]], Sy.sn_rowptr[sn + 1] - Sy.sn_rowptr[sn]);
            wave_mmin[Sy.sn_level[sn]] = std::min(wave_mmin[Sy.sn_level[sn]], Sy.sn_rowptr[sn + 1] - Sy.sn_rowptr[sn]);
        }
        chain_maxc = (int)std::max(1ll, knob_int("chain_solve_maxc", chain_maxc));
        if (!Sy.solve_group && chain_solve) {
            auto Kc = [&](int sn) { return Sy.sn_colptr[sn + 1] - Sy.sn_colptr[sn]; };
            auto Mr = [&](int sn) { return Sy.sn_rowptr[sn + 1] - Sy.sn_rowptr[sn]; };
            auto nchild = [&](int sn) { return Sy.child_ptr[sn + 1] - Sy.child_ptr[sn]; };
            auto cnt = [&](int l) { return Sy.level_ptr[(size_t)l * FC_COUNT + FC_COUNT] - Sy.level_ptr[(size_t)l * FC_COUNT]; };
            // multi-GPU: only runs of >= 4 levels made of pure links of the replicated top (nothing to gather inside the launch: the
            // distributed sweeps exchange the joins between launches)
            auto pure = [&](int sn) { return Sy.sn_class[sn] == FC_BIG && Sy.alias_child[sn] >= 0 && nchild(sn) == 1 && Kc(sn) <= 64; };
            std::vector<char> lvok(Sy.num_levels, 0);
            for (int lv = 0; lv < Sy.num_levels; ++lv) {
                const int a = Sy.level_ptr[(size_t)lv * FC_COUNT], b = Sy.level_ptr[(size_t)lv * FC_COUNT + FC_COUNT];
                bool ok = b > a && b - a <= chain_maxc;
                for (int q = a; q < b && ok; ++q) {
                    const int sn = Sy.level_sn[q];
                    if (multi) ok = pure(sn) && Sy.sn_level[Sy.alias_child[sn]] == lv - 1 && Sy.sn_owner[sn] < 0 && Sy.sn_gdepth[sn] == 0 && Sy.sn_gsz[sn] >= opt.nranks;
                    else       ok = Kc(sn) <= 64;          // (any class: to the sweeps a small front is a one-link chain like any other)
                }
                lvok[lv] = ok ? 1 : 0;
            }
            std::vector<int> chain_of(Sy.num_sn, -1);
            in_seg.assign(Sy.num_sn, 0);
            for (int lv = 0; lv < Sy.num_levels; ) {
                if (!lvok[lv]) { ++lv; continue; }
                int e = lv;
                while (e + 1 < Sy.num_levels && lvok[e + 1] && (!multi || cnt(e + 1) == cnt(lv))) ++e;
                if (e - lv + 1 >= (multi ? 4 : 2)) {
                    ChainSeg sg{lv, e, (int)chd.size(), 0, 0, 0, 0, (int)wgf.size(), (int)wgb.size()};
                    // chains, in the order of their first links' levels
                    const size_t chl0 = chl.size();
                    for (int l = lv; l <= e; ++l)
                        for (int q = Sy.level_ptr[(size_t)l * FC_COUNT]; q < Sy.level_ptr[(size_t)l * FC_COUNT + FC_COUNT]; ++q) {
                            const int sn = Sy.level_sn[q], ac = Sy.alias_child[sn];
                            in_seg[sn] = 1;
                            ChainLink L{}; L.panel_off = PO(sn); L.minv_off = Sy.minv_off[sn]; L.c0 = Sy.sn_colptr[sn]; L.k = Kc(sn); L.ldp = Sy.sn_ldp[sn];
                            L.s = sn; L.r0 = Sy.sn_rowptr[sn];
                            const int cprev = (ac >= 0 && nchild(sn) == 1 && Sy.sn_level[ac] >= lv) ? chain_of[ac] : -1;
                            if (cprev >= 0) {          // next link of its child's chain (the child is that chain's last link so far)
                                ChainDesc& D = chd[cprev];
                                L.koff = D.ktot; D.ktot += L.k; D.nlinks++; D.tail = Mr(sn) - L.k; L.fi = cprev;
                                chain_of[sn] = cprev;
                            } else {
                                ChainDesc D{}; D.cvb = Sy.cv_off[sn]; D.nlinks = 1; D.ktot = L.k; D.tail = Mr(sn) - L.k; D.ch0 = Sy.child_ptr[sn]; D.ch1 = Sy.child_ptr[sn + 1];
                                D.alias0 = ac >= 0 ? 1 : 0; D.s0 = sn;
                                L.koff = 0; L.fi = (int)chd.size();
                                chain_of[sn] = (int)chd.size(); chd.push_back(D);
                            }
                            chl.push_back(L);
                        }
                    sg.ndesc = (int)chd.size() - sg.desc0;
                    {   // flatten: links chain by chain, bottom link first
                        const size_t first = chl0;
                        std::vector<ChainLink> part(chl.begin() + first, chl.end());
                        std::stable_sort(part.begin(), part.end(), [](const ChainLink& x, const ChainLink& y) { return x.fi < y.fi; });
                        std::copy(part.begin(), part.end(), chl.begin() + first);
                        int cur = -1;
                        for (size_t t = first; t < chl.size(); ++t) { if (chl[t].fi != cur) { cur = chl[t].fi; chd[cur].link0 = (int)t; } }
                        for (size_t t = first; t < chl.size(); ++t) chl[t].fi = (int)t;
                    }
                    // workgroups: forward in chain order (children's chains first), backward in reverse (parents first)
                    for (int d = sg.desc0; d < sg.desc0 + sg.ndesc; ++d) {
                        ChainDesc& D = chd[d];
                        const int nt = (D.tail + 63) / 64;
                        D.wg0f = sg.nwg_f; D.tf0 = ntailflags; ntailflags += nt;
                        for (int w = 0; w < D.nlinks + nt; ++w) wgf.push_back(d);
                        sg.nwg_f += D.nlinks + nt; sg.maxtail = std::max(sg.maxtail, D.tail);
                    }
                    for (int d = sg.desc0 + sg.ndesc - 1; d >= sg.desc0; --d) {      // per chain: its dot workgroups (256 rows beyond the chain x one link each), then its links
                        ChainDesc& D = chd[d];
                        const int nw = D.nlinks * ((D.tail + 255) / 256 + 1);
                        D.wg0b = sg.nwg_b; sg.nwg_b += nw;
                        D.dot0 = ndots; ndots += D.nlinks * ((D.tail + 255) / 256);
                        for (int w = 0; w < nw; ++w) wgb.push_back(d);
                    }
                    // what a chain waits for: forward, the rows beyond each child chain of its first link (all of them before anything is
                    // gathered); backward, the bottom link of the chain its parent lives in
                    for (int d = sg.desc0; d < sg.desc0 + sg.ndesc; ++d) {
                        ChainDesc& D = chd[d];
                        D.gw0 = (int)chwait.size();
                        bool gathers = false;
                        for (int q = D.ch0; q < D.ch1; ++q) {
                            const int c = Sy.child_idx[q];
                            if (c != Sy.alias_child[D.s0]) gathers = true;
                            if (Sy.sn_level[c] < lv || chain_of[c] < 0) continue;
                            const ChainDesc& X = chd[chain_of[c]];
                            for (int b = 0; b < (X.tail + 63) / 64; ++b) chwait.push_back(X.tf0 + b);
                        }
                        D.gw1 = (int)chwait.size();
                        // 0: the vector is in place (in-place link of a front below the segment), 1: fresh vector, nothing to gather, 2: gather step
                        D.init = (gathers || D.gw1 > D.gw0) ? 2 : (D.alias0 ? 0 : 1);
                        if (multi && D.init != 0) { fprintf(stderr, "[mi355x_kkt] internal: chain segment with a gather in a distributed schedule\n"); return false; }
                        const int last = chl[D.link0 + D.nlinks - 1].s, par = Sy.sn_parent[last];
                        D.pw0 = (int)chwait.size();
                        if (par >= 0 && Sy.sn_level[par] <= e && chain_of[par] >= 0) { const ChainDesc& P = chd[chain_of[par]]; for (int t = 0; t < P.nlinks; ++t) chwait.push_back(P.link0 + t); }
                        D.pw1 = (int)chwait.size();
                    }
                    seg_at_lv0[lv] = seg_at_lv1[e] = (int)chain_segs.size(); chain_segs.push_back(sg);
                }
                lv = e + 1;
            }
            if (opt.verbose) { int nl = 0; for (auto& sg : chain_segs) nl += sg.lv1 - sg.lv0 + 1; fprintf(stderr, "[mi355x_kkt] data-flow solve sweeps: %d segments covering %d of %d levels, %d chains, %d links\n", (int)chain_segs.size(), nl, Sy.num_levels, (int)chd.size(), (int)chl.size()); }
        }
        if (!upload(wgf, &V.chwg_f) || !upload(wgb, &V.chwg_b) || !upload(chwait, &V.chwait)) return false;
        V.strace = nullptr; V.strace_b = (int)wgf.size(); strace_n = 0;
        if (knob_trace("solve") && !wgf.empty()) {
            strace_n = 4 * (wgf.size() + wgb.size());
            if (!dalloc(&V.strace, strace_n)) return false;
            strace_desc.clear();
            for (size_t i = 0; i < wgf.size(); ++i) { const ChainDesc& D = chd[wgf[i]]; strace_desc.push_back({wgf[i], (int)i - D.wg0f - chain_segs[0].wgf0 * 0, D.nlinks, D.tail}); }
            for (size_t i = 0; i < wgb.size(); ++i) { const ChainDesc& D = chd[wgb[i]]; strace_desc.push_back({wgb[i], (int)i - D.wg0b - D.nlinks * ((D.tail + 255) / 256), D.nlinks, D.tail}); }      // (dot workgroups: negative)
        }
        n_sflag_dot = (size_t)std::max(ndots, 1) * FLAG_STRIDE; n_dpart = (size_t)std::max(ndots, 1) * 64; n_sflag_t = (size_t)std::max(ntailflags, 1) * FLAG_STRIDE; n_sflag_b = std::max<size_t>(chl.size(), 1) * FLAG_STRIDE;
        if (!dalloc(&V.sflag_dot, n_sflag_dot) || !dalloc(&V.dpart, n_dpart) || !dalloc(&V.sflag_t, n_sflag_t) || !dalloc(&V.sflag_b, n_sflag_b)) return false;
        {   // tagged solution entries of the segments' columns (indexed by column; + 64: a wavefront polls 64 entries from a link's first column)
            const size_t nt = chl.empty() ? 1 : (size_t)Sy.n + 64;
            n_tag = nt;
            if (!dalloc(&V.ytag, nt) || !dalloc(&V.xtag, nt)) return false;
        }
        if (!upload(chl, &V.chlink) || !upload(chd, &V.chdesc)) return false;
        std::vector<int> bigidx_of(Sy.num_sn, 0);
        { int nbig = 0; for (int sn = 0; sn < Sy.num_sn; ++sn) if (Sy.sn_class[sn] == FC_BIG) bigidx_of[sn] = nbig++; }
        // chain-group tables: for every BIG front the links of its group up to and including itself
        std::vector<GroupLink> gt;
        std::vector<int> gbase_of(Sy.num_sn, 0), gcols_of(Sy.num_sn, 0);
        for (int sn = 0; sn < Sy.num_sn; ++sn) {
            gcols_of[sn] = Sy.sn_colptr[sn + 1] - Sy.sn_colptr[sn];
            if (Sy.sn_class[sn] != FC_BIG) continue;
            std::vector<int> links(Sy.grp_pos[sn] + 1);
            int cur = sn;
            for (int j = Sy.grp_pos[sn]; j >= 0; --j) { links[j] = cur; if (j > 0) cur = Sy.alias_child[cur]; }
            gbase_of[sn] = (int)gt.size(); gcols_of[sn] = 0;
            for (int l : links) {
                GroupLink G;
                G.panel_off = PO(l); G.wb = Sy.wb_off[l]; G.minv_off = Sy.minv_off[l]; G.cv = Sy.cv_off[l]; G.tr = troff[l];
                G.c0 = Sy.sn_colptr[l]; G.k = Sy.sn_colptr[l + 1] - G.c0; G.r0 = Sy.sn_rowptr[l]; G.m = Sy.sn_rowptr[l + 1] - G.r0;
                G.ldp = Sy.sn_ldp[l]; G.ch0 = Sy.child_ptr[l]; G.ch1 = Sy.child_ptr[l + 1]; G.alias = Sy.alias_child[l] >= 0 ? 1 : 0;
                G.t_off = CO(l); G.ldt = Sy.sn_ldt[l]; G.s = l; G.aq0 = Sy.acolptr[G.c0]; G.aq1 = Sy.acolptr[G.c0 + G.k]; G.bigidx = bigidx_of[l];
                G.selfasm = ((!multi || aoff[l] < 0) && !knob_disabled("selfasm") && Sy.alias_child[l] >= 0 && Sy.child_ptr[l + 1] - Sy.child_ptr[l] == 1) ? 1 : 0;
                gt.push_back(G); gcols_of[sn] += G.k;
            }
        }
        // look-ahead candidates: group-last BIG fronts (>= la_min_nt = 8 tile rows; 12 before part 1 had its 64 x 64 tiles) whose chain continues with a PURE next group (links
        // whose only child is the chain child: nothing but the chain itself writes into the front before the next full update)
        std::vector<char> split_of(Sy.num_sn, 0);
        la_tiles1.assign(Sy.num_levels, 0); la_tiles2.assign(Sy.num_levels, 0); la_full.assign(Sy.num_levels, 0);
        {
            std::vector<int> alias_parent(Sy.num_sn, -1);
            for (int sn = 0; sn < Sy.num_sn; ++sn) if (Sy.alias_child[sn] >= 0) alias_parent[Sy.alias_child[sn]] = sn;
            for (int sn = 0; sn < Sy.num_sn; ++sn) {
                if (Sy.sn_class[sn] != FC_BIG || Sy.grp_rem[sn] != 0) continue;
                const int lv = Sy.sn_level[sn];
                la_full[lv] = 1;
                const int mu = (Sy.sn_rowptr[sn + 1] - Sy.sn_rowptr[sn]) - (Sy.sn_colptr[sn + 1] - Sy.sn_colptr[sn]);
                const int nt = (mu + 127) / 128;
                bool ok = lookahead && !multi && nt >= la_min_nt && alias_parent[sn] >= 0;
                for (int p = alias_parent[sn]; ok && p >= 0; p = alias_parent[p]) {
                    if (Sy.child_ptr[p + 1] - Sy.child_ptr[p] != 1) ok = false;
                    if (Sy.grp_rem[p] == 0) break;
                }
                split_of[sn] = ok ? 1 : 0;
                la_tiles1[lv] = std::max(la_tiles1[lv], ok ? 2 * nt - 1 : tri_tiles(nt));
                if (ok) la_tiles2[lv] = std::max(la_tiles2[lv], ((nt - 2) * (nt - 1) / 2 + 7) / 8 * 8);
            }
            for (int sn = 0; sn < Sy.num_sn; ++sn) if (Sy.sn_class[sn] == FC_BIG && Sy.grp_rem[sn] != 0)
                la_tiles1[Sy.sn_level[sn]] = std::max(la_tiles1[Sy.sn_level[sn]], schur_tiles(Sy, sn));
            if (opt.verbose) {
                int nsplit = 0, nfull = 0, nimpure = 0, nsmall = 0, nend = 0; long long t2 = 0;
                for (int sn = 0; sn < Sy.num_sn; ++sn) if (Sy.sn_class[sn] == FC_BIG && Sy.grp_rem[sn] == 0) {
                    ++nfull; nsplit += split_of[sn];
                    const int mu = (Sy.sn_rowptr[sn + 1] - Sy.sn_rowptr[sn]) - (Sy.sn_colptr[sn + 1] - Sy.sn_colptr[sn]);
                    const int nt = (mu + 127) / 128;
                    if (split_of[sn]) t2 += (long long)(nt - 2) * (nt - 1) / 2;
                    else if (nt < la_min_nt) ++nsmall; else if (alias_parent[sn] < 0) ++nend; else ++nimpure;
                }
                fprintf(stderr, "[mi355x_kkt] look-ahead: %d of %d group-end updates split (%lld part-2 tiles); not split: %d small, %d chain ends, %d impure next group\n",
                        nsplit, nfull, t2, nsmall, nend, nimpure);
            }
            // look-ahead costs the graph replay (see factor()): only worth it when a good part of the flops is in split updates
            long long la_total = 0;
            for (int sn = 0; sn < Sy.num_sn; ++sn) if (split_of[sn]) {
                const int mu = (Sy.sn_rowptr[sn + 1] - Sy.sn_rowptr[sn]) - (Sy.sn_colptr[sn + 1] - Sy.sn_colptr[sn]);
                const int nt = (mu + 127) / 128; la_total += (long long)(nt - 2) * (nt - 1) / 2;
            }
            const long long la_gate = knob_int("la_min_tiles", 4000);     // (tests force 0)
            if (la_total < la_gate || la_total == 0) { std::fill(split_of.begin(), split_of.end(), 0); std::fill(la_tiles2.begin(), la_tiles2.end(), 0);
                for (int lv = 0; lv < Sy.num_levels; ++lv) la_tiles1[lv] = 0; }
            la_any = la_total >= la_gate && la_total > 0;
            la_evA.assign(Sy.num_levels, nullptr); la_evB.assign(Sy.num_levels, nullptr);
            for (int lv = 0; lv < Sy.num_levels; ++lv) if (la_tiles2[lv] > 0) {
                HIPCHK(hipEventCreateWithFlags(&la_evA[lv], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&la_evB[lv], hipEventDisableTiming)); }
        }
        // XCD-aware tile orders for the large (>= 12 tile rows) full updates, one table per triangle size
        std::vector<int> ttab_of(Sy.num_sn, -1), ttab2_of(Sy.num_sn, -1), tile_tab;
        {
            std::map<int, int> tab_at;
            auto table_for = [&](int n) {
                auto it = tab_at.find(n);
                if (it != tab_at.end()) return it->second;
                const int at = (int)tile_tab.size(), S8 = 8, nst = (n + S8 - 1) / S8;
                for (int I = 0; I < nst; ++I) for (int J = 0; J <= I; ++J)
                    for (int a = 0; a < S8; ++a) for (int b = 0; b < S8; ++b) {
                        const int ti = I * S8 + a, tc = J * S8 + b;
                        if (ti < n && tc <= ti) tile_tab.push_back((ti << 16) | tc);
                    }
                tab_at[n] = at; return at;
            };
            const bool xcd_aware = !knob_disabled("xcd_tiles");
            for (int sn = 0; sn < Sy.num_sn; ++sn) if (xcd_aware && Sy.sn_class[sn] == FC_BIG && Sy.grp_rem[sn] == 0) {
                const int mu = (Sy.sn_rowptr[sn + 1] - Sy.sn_rowptr[sn]) - (Sy.sn_colptr[sn + 1] - Sy.sn_colptr[sn]);
                const int nt = (mu + 127) / 128;
                if (nt < 12) continue;
                ttab_of[sn] = table_for(nt);
                if (split_of[sn]) ttab2_of[sn] = table_for(nt - 2);
            }
        }
        if (!upload(tile_tab, &V.tile_tab)) return false;
        const bool selfasm_on = !knob_disabled("selfasm");
        lv_asm_skip.assign(Sy.num_levels, 0);
        if (!multi && selfasm_on)
            for (int lv = 0; lv < Sy.num_levels; ++lv) {
                bool all = true; int cnt = 0;
                for (int q = Sy.level_ptr[(size_t)lv * FC_COUNT + FC_BIG]; q < Sy.level_ptr[(size_t)lv * FC_COUNT + FC_BIG + 1]; ++q) {
                    const int sn = Sy.level_sn[q]; ++cnt;
                    if (!(Sy.alias_child[sn] >= 0 && Sy.child_ptr[sn + 1] - Sy.child_ptr[sn] == 1)) all = false;
                }
                lv_asm_skip[lv] = (cnt > 0 && all) ? 1 : 0;
            }
        // levels whose big fronts are ALL chain links that are not the last of their group: the (narrow) trailing updates ride in
        // the fused pivot-block + panel-solve launch
        lv_narrow_tiles.assign(Sy.num_levels, 0);
        if (!multi && !knob_disabled("fuse_upd"))
            for (int lv = 0; lv < Sy.num_levels; ++lv) {
                bool all = true; int cnt = 0, tl = 0;
                for (int q = Sy.level_ptr[(size_t)lv * FC_COUNT + FC_BIG]; q < Sy.level_ptr[(size_t)lv * FC_COUNT + FC_BIG + 1]; ++q) {
                    const int sn = Sy.level_sn[q]; ++cnt;
                    if (Sy.grp_rem[sn] <= 0) all = false;
                    tl = std::max(tl, schur_tiles64(Sy, sn));
                }
                lv_narrow_tiles[lv] = (cnt > 0 && all) ? tl : 0;
            }
        // ---- grouped schedule: every chain group is factored at the level of its first link (k_grp_fused + update) ----
        grouped = Sy.maxsupernode <= 64 && selfasm_on && !knob_disabled("grouped");
        {
            auto order_of = [&](int sn) { return Sy.sn_rowptr[sn + 1] - Sy.sn_rowptr[sn]; };
            auto cols_of = [&](int sn) { return Sy.sn_colptr[sn + 1] - Sy.sn_colptr[sn]; };
            // which: 0 = every front (one GPU), 1 = the rank's own subtrees, 2 + d = the replicated fronts of exchange step d held by this rank (an
            // in-place chain never crosses an ownership boundary -- symbolic.cpp only aliases fronts of one owner and one range of ranks -- so neither does a group)
            auto build_groups = [&](GrpSched& G, int which) -> bool {
                for (auto* v : {&G.g0, &G.g1, &G.split, &G.nrb, &G.tiles64, &G.tiles, &G.la1, &G.la2, &G.p1t, &G.la3, &G.nsplit}) v->assign(Sy.num_levels, 0);
                G.evA.assign(Sy.num_levels, nullptr); G.evB.assign(Sy.num_levels, nullptr);
                if (!grouped) return true;
                std::vector<std::vector<int>> at(Sy.num_levels);
                for (int sn = 0; sn < Sy.num_sn; ++sn) if (Sy.sn_class[sn] == FC_BIG && Sy.grp_rem[sn] == 0) {
                    if (which == 1 && Sy.sn_owner[sn] != opt.rank) continue;
                    if (which >= 2 && !(Sy.sn_owner[sn] < 0 && Sy.sn_glo[sn] <= opt.rank && opt.rank < Sy.sn_glo[sn] + Sy.sn_gsz[sn] && Sy.sn_gdepth[sn] == which - 2)) continue;
                    int first = sn;
                    for (int j = Sy.grp_pos[sn]; j > 0; --j) first = Sy.alias_child[first];
                    if (Sy.sn_level[first] >= Sy.grp_cut_level) at[Sy.sn_level[first]].push_back(sn);      // (groups do not straddle the cut: symbolic.cpp)
                }
                int ng = 0;
                for (int lv = 0; lv < Sy.num_levels; ++lv) {
                    std::stable_sort(at[lv].begin(), at[lv].end(), [&](int a, int b) { return order_of(a) < order_of(b); });
                    G.g0[lv] = (int)lvl_list.size();
                    int nsmall = 0;
                    for (int sn : at[lv]) {
                        lvl_list.push_back(sn); ++ng;
                        const int mu = order_of(sn) - cols_of(sn), nt = (mu + 127) / 128;
                        G.nrb[lv] = std::max(G.nrb[lv], (mu + 63) / 64);
                        if (order_of(sn) <= 1024) { ++nsmall; G.tiles64[lv] = std::max(G.tiles64[lv], schur_tiles64(Sy, sn)); }
                        else {
                            G.tiles[lv] = std::max(G.tiles[lv], schur_tiles(Sy, sn));
                            G.la1[lv] = std::max(G.la1[lv], split_of[sn] ? 2 * nt - 1 : tri_tiles(nt));
                            if (split_of[sn]) G.la2[lv] = std::max(G.la2[lv], ((nt - 2) * (nt - 1) / 2 + 7) / 8 * 8);
                            if (split_of[sn]) { const int n64 = (mu + 63) / 64; int c = 0; for (int tc = 0; tc < 4 && tc < n64; ++tc) c += n64 - tc; G.p1t[lv] = std::max(G.p1t[lv], c); ++G.nsplit[lv]; }
                            else G.la3[lv] = std::max(G.la3[lv], tri_tiles(nt));
                        }
                    }
                    G.split[lv] = nsmall; G.g1[lv] = (int)lvl_list.size();
                    if (G.la2[lv] > 0) { HIPCHK(hipEventCreateWithFlags(&G.evA[lv], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&G.evB[lv], hipEventDisableTiming)); }
                }
                if (opt.verbose) fprintf(stderr, "[mi355x_kkt] grouped schedule%s: %d chain groups factored in one launch each (tree levels >= %d)\n",
                                         which == 0 ? "" : (which == 1 ? " (own subtrees)" : " (replicated fronts of one exchange step)"), ng, Sy.grp_cut_level);
                return true;
            };
            if (!multi) { if (!build_groups(gs_single, 0)) return false; }
            else { if (!build_groups(gs_local, 1)) return false;
                   gs_stage.assign(ndepth, GrpSched());
                   for (int d = 0; d < ndepth; ++d) if (!build_groups(gs_stage[d], 2 + d)) return false; }
        }
        asm_fast_ok.assign(lvl_list.size(), 0); h_asmcut.assign(lvl_list.size(), 0);
        for (size_t q = 0; q < lvl_list.size(); ++q) {
            const int sn = lvl_list[q];
            int nch = 0;
            for (int c = Sy.child_ptr[sn]; c < Sy.child_ptr[sn + 1]; ++c) if (Sy.child_idx[c] != Sy.alias_child[sn]) ++nch;
            const bool chain_only = Sy.alias_child[sn] >= 0 && nch == 0;      // (pure in-place link: nothing to assemble either way)
            asm_fast_ok[q] = chain_only ? 1 : (Sy.alias_child[sn] < 0 ? (nch <= 6 ? std::max(nch, 1) : 0) : 0);      // (the number of children the front brings to k_big_assemble2's row maps -- a front with more than ASM_MAXCH falls back to the column routine inside the kernel; 0: not for that kernel)
        }
        // tfuse: the contribution block of a front is formed by its trailing update (T = sum of the children's contributions - L21 W21^T, written once) instead of
        // being assembled, read back and written again.  A front that is a unit of its own -- assembled (not in place on a child), its update one launch of
        // k_big_schur64 / k_big_schur -- or (round 5) a whole CHAIN GROUP whose first link is assembled: that link's assembly stops at the group's columns
        // (asmcut: the panels of all the group's links), the trailing block of the LAST link is formed by the group-end update out of the FIRST link's children.
        // MI355X_KKT_TFUSE_SMALL: the scope of round 4 (fronts of order <= 1024 below the grouped top that are no chain group).
        std::vector<char> tfuse_of(Sy.num_sn, 0);
        std::vector<int> asmcut_of(Sy.num_sn, 0);
        ntfuse = 0;
        if (!multi && !knob_disabled("tfuse")) {
            const bool wide = true;
            for (int sn = 0; sn < Sy.num_sn; ++sn) {
                if (Sy.sn_class[sn] != FC_BIG || Sy.grp_rem[sn] != 0) continue;      // (the last link of its group, or a front of its own)
                int first = sn;
                for (int j = 0; j < Sy.grp_pos[sn]; ++j) first = Sy.alias_child[first];
                const int m = Sy.sn_rowptr[sn + 1] - Sy.sn_rowptr[sn], k = Sy.sn_colptr[sn + 1] - Sy.sn_colptr[sn], nch = Sy.child_ptr[first + 1] - Sy.child_ptr[first];
                if (Sy.alias_child[first] >= 0 || m <= k) continue;
                if (!wide) {
                    if (m > 1024 || nch > 16 || Sy.grp_pos[sn] != 0) continue;
                    if (grouped && Sy.sn_level[sn] >= Sy.grp_cut_level) continue;
                } else if (nch > 96) continue;
                tfuse_of[sn] = 1; ++ntfuse;
                asmcut_of[first] = (Sy.sn_rowptr[first + 1] - Sy.sn_rowptr[first]) - (m - k);      // (= k of a front of its own, the group's columns otherwise)
            }
        }
        if (opt.verbose) fprintf(stderr, "[mi355x_kkt] contribution blocks formed by their update (not assembled): %d of %d big fronts\n", ntfuse, Sy.num_big);
        std::vector<FrontMeta> fm(lvl_list.size());
        for (size_t q = 0; q < lvl_list.size(); ++q) {
            const int sn = lvl_list[q];
            FrontMeta& M = fm[q];
            M.tfuse = tfuse_of[sn]; M.asmcut = asmcut_of[sn]; h_asmcut[q] = M.asmcut;      // (-1 below: a pure in-place link, nothing to assemble)
            M.s = sn; M.c0 = Sy.sn_colptr[sn]; M.k = Sy.sn_colptr[sn + 1] - M.c0; M.r0 = Sy.sn_rowptr[sn]; M.m = Sy.sn_rowptr[sn + 1] - M.r0;
            M.aq0 = Sy.acolptr[M.c0]; M.aq1 = Sy.acolptr[M.c0 + M.k]; M.ch0 = Sy.child_ptr[sn]; M.ch1 = Sy.child_ptr[sn + 1]; M.alias = Sy.alias_child[sn] >= 0 ? 1 : 0;
            M.ldp = Sy.sn_ldp[sn]; M.ldt = Sy.sn_ldt[sn];
            M.panel_off = PO(sn); M.cb_off = CO(sn); M.minv_off = Sy.minv_off[sn];
            M.cv = Sy.cv_off[sn]; M.wb = Sy.wb_off[sn]; M.gpart = Sy.gpart_off[sn];
            M.gbase = gbase_of[sn]; M.gpos = Sy.grp_pos[sn]; M.grem = Sy.grp_rem[sn]; M.gcols = gcols_of[sn]; M.split = split_of[sn]; M.ttab = ttab_of[sn]; M.ttab2 = ttab2_of[sn];
            // (multi-GPU: not for a front at a subtree join -- its square comes out of the all-reduced arena)
            M.selfasm = ((!multi || aoff[sn] < 0) && selfasm_on && Sy.sn_class[sn] == FC_BIG && Sy.alias_child[sn] >= 0 && Sy.child_ptr[sn + 1] - Sy.child_ptr[sn] == 1) ? 1 : 0; M.bigidx = bigidx_of[sn];
            {   // 1: in-place chain link whose only child is the chain child, 2: no children at all => the fused forward kernel applies
                const int nch = Sy.child_ptr[sn + 1] - Sy.child_ptr[sn];
                M.solo = (Sy.alias_child[sn] >= 0 && nch == 1) ? 1 : ((Sy.alias_child[sn] < 0 && nch == 0) ? 2 : 0);
            }
            if (M.selfasm && !M.asmcut) h_asmcut[q] = -1;
            if (q < solve_entry.size() && solve_entry[q] && !Sy.solve_group) {     // per-link solves: every front is its own unit
                M.gbase += M.gpos; M.gpos = 0; M.grem = 0; M.gcols = M.k;
            }
        }
        // ChildMeta::owner as the kernels read it: -1 = assembled directly by the parent (same owner / same range of ranks); otherwise the child
        // reaches its (replicated) parent through the arena and the value says who reports it and in which exchange step: reporting rank
        // + nranks * (0 for an owned subtree root, 1 + depth for a front of a sub-range) -- the `who` of k_arena_assemble / k_top_rhs_assemble
        auto child_code = [&](int ch) {
            const int pa = Sy.sn_parent[ch], P = std::max(1, opt.nranks);
            if (!multi || pa < 0 || Sy.sn_owner[pa] >= 0) return Sy.sn_owner[ch];
            if (Sy.sn_owner[ch] >= 0) return Sy.sn_owner[ch];
            if (Sy.sn_glo[ch] == Sy.sn_glo[pa] && Sy.sn_gsz[ch] == Sy.sn_gsz[pa]) return -1;
            return Sy.sn_glo[ch] + P * (1 + Sy.sn_gdepth[ch]);
        };
        std::vector<ChildMeta> cm(Sy.child_idx.size());
        for (size_t q = 0; q < cm.size(); ++q) {
            const int ch = Sy.child_idx[q]; const int kc = Sy.sn_colptr[ch + 1] - Sy.sn_colptr[ch];
            cm[q].ch = ch; cm[q].relbase = Sy.sn_rowptr[ch] + kc; cm[q].mc = Sy.sn_rowptr[ch + 1] - cm[q].relbase;
            cm[q].owner = child_code(ch); cm[q].cb_off = CO(ch); cm[q].ldt = Sy.sn_ldt[ch]; cm[q].aliased = 0; cm[q].cvbase = Sy.cv_off[ch] + kc;
            cm[q].rlo = cm[q].mc > 0 ? Sy.rel[cm[q].relbase] : (1 << 30); cm[q].rhi = cm[q].mc > 0 ? Sy.rel[cm[q].relbase + cm[q].mc - 1] : -1;      // (rel is ascending)
        }
        for (int sn = 0; sn < Sy.num_sn; ++sn) if (Sy.alias_child[sn] >= 0)
            for (int q = Sy.child_ptr[sn]; q < Sy.child_ptr[sn + 1]; ++q) if (Sy.child_idx[q] == Sy.alias_child[sn]) cm[q].aliased = 1;
        // inverse relative indices for the children of BIG parents (k_big_assemble: one load instead of a binary search per column)
        std::vector<int> relinv;
        for (int sn = 0; sn < Sy.num_sn; ++sn) {
            for (int q = Sy.child_ptr[sn]; q < Sy.child_ptr[sn + 1]; ++q) cm[q].inv = 0;
            if (Sy.sn_class[sn] != FC_BIG && !(sn < (int)in_seg.size() && in_seg[sn])) continue;
            const int mp = Sy.sn_rowptr[sn + 1] - Sy.sn_rowptr[sn];
            for (int q = Sy.child_ptr[sn]; q < Sy.child_ptr[sn + 1]; ++q) {
                if (cm[q].aliased) continue;
                cm[q].inv = (long long)relinv.size();
                relinv.resize(relinv.size() + mp, -1);
                int* inv = relinv.data() + cm[q].inv;
                for (int a = 0; a < cm[q].mc; ++a) inv[Sy.rel[cm[q].relbase + a]] = a;
            }
        }
        if (!upload(relinv, &V.relinv)) return false;
        {   // per big front: room for the 16 x 16 diagonal-block inverses of its L11 (ISG_STRIDE doubles each, slot FrontMeta::bigidx)
            long long nbig = 0;
            for (int sn = 0; sn < Sy.num_sn; ++sn) if (Sy.sn_class[sn] == FC_BIG) ++nbig;
            if (!dalloc(&V.isg, (size_t)std::max<long long>(nbig * ISG_STRIDE, 1)) || !dalloc(&V.hasis, Sy.num_sn)) return false;
        }
        lap("host-side schedules and tables");
        if (!upload(fm, &V.fmeta) || !upload(cm, &V.cmeta) || !upload(gt, &V.gtab)) return false;
        {   // leaf chains (k_leaf_chain): the levels below lc_levels hold nothing but fronts of order <= 16 with at most one child -- every such front is a
            // link of the chain that starts at its leaf
            lc_levels = 0; lc_nchains = 0;
            std::vector<int> lcp, lcf; std::vector<LeafLink> lcl;
            if (!multi && V.fastpiv && !knob_disabled("leafchain")) {
                int L = 0;
                for (; L < Sy.num_levels && L < 16; ++L) {
                    bool okl = true; int cnt = 0;
                    for (int fc = 0; fc < FC_COUNT && okl; ++fc) {
                        const int b0 = Sy.level_ptr[(size_t)L * FC_COUNT + fc], b1 = Sy.level_ptr[(size_t)L * FC_COUNT + fc + 1];
                        if (fc != FC_WAVE) { if (b1 > b0) okl = false; continue; }
                        for (int q = b0; q < b1 && okl; ++q) {
                            const int sn = lvl_list[q]; ++cnt;
                            if (Sy.sn_rowptr[sn + 1] - Sy.sn_rowptr[sn] > 16 || Sy.child_ptr[sn + 1] - Sy.child_ptr[sn] > 1) okl = false;
                        }
                    }
                    if (!okl || cnt == 0) break;
                }
                if (L >= 2) {
                    lc_levels = L;
                    std::vector<int> fmw(Sy.num_sn, -1);
                    for (int lv = 0; lv < L; ++lv) for (int q = Sy.level_ptr[(size_t)lv * FC_COUNT + FC_WAVE]; q < Sy.level_ptr[(size_t)lv * FC_COUNT + FC_WAVE + 1]; ++q) fmw[lvl_list[q]] = q;
                    lcp.push_back(0);
                    for (int q = Sy.level_ptr[(size_t)0 * FC_COUNT + FC_WAVE]; q < Sy.level_ptr[(size_t)0 * FC_COUNT + FC_WAVE + 1]; ++q) {
                        for (int cur = lvl_list[q]; cur >= 0 && Sy.sn_level[cur] < L; cur = Sy.sn_parent[cur]) lcf.push_back(fmw[cur]);
                        lcp.push_back((int)lcf.size());
                    }
                    lc_nchains = (int)lcp.size() - 1;
                    if ((int)lcf.size() != Sy.level_ptr[(size_t)(L - 1) * FC_COUNT + FC_WAVE + 1] - Sy.level_ptr[(size_t)0 * FC_COUNT + FC_WAVE]) { lc_levels = 0; lc_nchains = 0; }   // (cannot happen: every front below L is on exactly one chain)
                    if (opt.verbose && lc_levels) fprintf(stderr, "[mi355x_kkt] leaf chains: %d chains over the bottom %d levels (%zu fronts) in one launch per sweep\n", lc_nchains, lc_levels, lcf.size());
                }
            }
            if (lcp.empty()) lcp.push_back(0);
            for (int q : lcf) {
                const FrontMeta& M = fm[q]; LeafLink K;
                K.s = M.s; K.c0 = M.c0; K.k = M.k; K.m = M.m; K.aq0 = M.aq0; K.aq1 = M.aq1; K.ldp = M.ldp; K.r0 = M.r0; K.pad = 0;
                K.relbase = M.r0 + M.k;                                  // (V.rel + relbase: where the front's update rows sit in its parent)
                K.panel_off = M.panel_off; K.minv_off = M.minv_off; K.cb_off = M.cb_off; K.cv = M.cv;
                lcl.push_back(K);
            }
            if (lcl.empty()) lcl.push_back(LeafLink());
            if (!upload(lcp, &V.lc_ptr) || !upload(lcl, &V.lc_link)) return false;
        }
        {   // k_front_df: maximal runs of >= 2 consecutive levels (above the leaf chains) whose fronts are all one-wavefront fronts
            df_runs.clear(); df_run_at.assign(Sy.num_levels, -1);
            std::vector<DfLevel> tab; std::vector<long long> cto(std::max(Sy.num_sn, 1), -1); std::vector<int> run_of(std::max(Sy.num_sn, 1), -1); long long cbt_len = 0;
            if (!multi && V.fastpiv && df_on) {
                auto pure = [&](int lv) {
                    const int w0 = Sy.level_ptr[(size_t)lv * FC_COUNT + FC_WAVE], w1 = Sy.level_ptr[(size_t)lv * FC_COUNT + FC_WAVE + 1];
                    return w1 > w0 && w1 - w0 == Sy.level_ptr[(size_t)lv * FC_COUNT + FC_COUNT] - Sy.level_ptr[(size_t)lv * FC_COUNT];
                };
                for (int lv = lc_levels; lv < Sy.num_levels; ) {
                    if (!pure(lv)) { ++lv; continue; }
                    int e = lv; while (e + 1 < Sy.num_levels && pure(e + 1)) ++e;
                    if (e > lv) {
                        DfRun R{lv, e, (int)tab.size(), e - lv + 1, 0};
                        for (int l = lv; l <= e; ++l) {
                            const int w0 = Sy.level_ptr[(size_t)l * FC_COUNT + FC_WAVE], w1 = Sy.level_ptr[(size_t)l * FC_COUNT + FC_WAVE + 1];
                            tab.push_back(DfLevel{w0, tiny16[l], w1 - w0, R.nq});
                            R.nq += (tiny16[l] + 3) / 4 + (w1 - w0 - tiny16[l]);
                            for (int q = w0; q < w1; ++q) run_of[lvl_list[q]] = (int)df_runs.size();
                        }
                        df_run_at[lv] = (int)df_runs.size(); df_runs.push_back(R);
                    }
                    lv = e + 1;
                }
                int per_cu = 0, ncu = 0;
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_front_df, 64, 0);
                (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
                df_grid_cap = std::max(1, per_cu) * std::max(1, ncu);      // every workgroup of the launch must be resident (they wait for each other)
                if (opt.verbose && !df_runs.empty()) { int nl = 0; for (auto& R : df_runs) nl += R.nlev; fprintf(stderr, "[mi355x_kkt] small fronts: %d runs covering %d of %d levels in one data-flow launch each (grid <= %d workgroups)\n", (int)df_runs.size(), nl, Sy.num_levels, df_grid_cap); }
            }
            // tagged contribution blocks: for every front of a run whose PARENT is a front of the same run
            for (int sn = 0; sn < Sy.num_sn; ++sn) {
                const int pa = Sy.sn_parent[sn];
                if (run_of[sn] < 0 || pa < 0 || run_of[pa] != run_of[sn]) continue;
                const int mu = (Sy.sn_rowptr[sn + 1] - Sy.sn_rowptr[sn]) - (Sy.sn_colptr[sn + 1] - Sy.sn_colptr[sn]);
                if (mu <= 0) continue;
                cto[sn] = cbt_len; cbt_len += mu <= 16 ? 256 : 1024;
            }
            if (tab.empty()) tab.push_back(DfLevel{0, 0, 0, 0});
            if (!upload(tab, &d_dftab)) return false;
            if (!upload(cto, &V.cbt_off)) return false;
            if (!dalloc(&V.cbt, (size_t)std::max<long long>(cbt_len, 1))) return false;
        }
        // the inertia / pivot counts are summed over the ranks: a replicated front is counted by the first rank of its range (-1 in this rank's view), -3 = not here
        std::vector<int> stat_owner(Sy.sn_owner.begin(), Sy.sn_owner.end());
        if (multi) for (int sn = 0; sn < Sy.num_sn; ++sn) if (Sy.sn_owner[sn] < 0) stat_owner[sn] = Sy.sn_glo[sn] == opt.rank ? -1 : -3;
        if (!upload(Sy.sn_colptr, &V.sn_colptr) || !upload(Sy.sn_rowptr, &V.sn_rowptr) || !upload(Sy.sn_rows, &V.sn_rows) ||
            !upload(Sy.rel, &V.rel) || !upload(Sy.child_ptr, &V.child_ptr) || !upload(Sy.child_idx, &V.child_idx) ||
            !upload(stat_owner, &V.sn_owner) || !upload(poff, &V.panel_off) || !upload(coff, &V.cb_off) || !upload(moff, &V.minv_off) ||
            !upload(Sy.acolptr, &V.acolptr) || !upload(Sy.apos, &V.apos) || !upload(Sy.arow, &V.arow) || !upload(Sy.acol, &V.acol) ||
            !upload(Sy.dup_ptr, &V.dup_ptr) || !upload(Sy.dup_src, &V.dup_src) || !upload(Sy.rslot_ptr, &V.rslot_ptr) || !upload(Sy.rslot_idx, &V.rslot_idx) || !upload(Sy.rslot_col, &V.rslot_col) || !upload(lvl_list, &V.level_sn) ||
            !upload(Sy.sn_parent, &V.sn_parent) || !upload(colown, &V.col_owner) || !upload(aoff, &V.arena_off) || !upload(troff, &V.top_rhs_off) ||
            !upload(Sy.perm, &V.perm)) return false;
        lap("uploads");
        double* tv = keep ? keep_tvals : nullptr;
        if (!tv && !dalloc(&tv, Sy.nnz_in)) return false;
        V.tvals = tv;
        V.rslot_len = (int)Sy.rslot_idx.size();
        if (!dalloc(&V.arv, Sy.rslot_idx.size()) || !dalloc(&V.aval, Sy.nnz_a) || !dalloc(&V.scale, Sy.n) || !dalloc(&V.scale2, Sy.n) || !dalloc(&V.rowmax, Sy.n) ||
            !dalloc(&V.wbuf, (size_t)Sy.wbuf_doubles) || !dalloc(&V.minv, (size_t)Sy.minv_doubles) ||
            !dalloc(&V.dinv, Sy.n) || !dalloc(&V.doff, Sy.n) || !dalloc(&V.ptype, Sy.n) || !dalloc(&V.lperm, Sy.n) ||
            !dalloc(&V.fstat, Sy.num_sn) || !dalloc(&V.xw, Sy.n) || !dalloc(&V.ybuf, Sy.n) || !dalloc(&V.zb, Sy.n) || !dalloc(&V.bw, Sy.n) || !dalloc(&V.xacc, Sy.n) || !dalloc(&V.cvec, (size_t)Sy.cvec_doubles) || !dalloc(&V.gpart, (size_t)Sy.gpart_doubles) ||
            !dalloc(&d_stats, 12) || !dalloc(&V.colfail, Sy.n) || !dalloc(&V.zpiv, Sy.n) || !dalloc(&V.cnorm, Sy.n) ||
            !dalloc(&V.sflag_d, Sy.num_sn) || !dalloc(&V.sflag_p, Sy.num_sn) || !dalloc(&V.sflag_s, 4 * (size_t)Sy.num_sn) || !dalloc(&V.tcnt, Sy.num_sn) || !dalloc(&V.apfail, Sy.num_sn) || !dalloc(&V.sepoch, 4)) return false;
        V.qstat = d_stats + 4;
        lap("device allocations");
        if (opt.scaling >= 3 && !d_user_scale) { HIPCHK(hipMalloc((void**)&d_user_scale, std::max<size_t>(Sy.n, 1) * sizeof(double))); allocs.push_back(d_user_scale); }
        else if (opt.scaling == 2 && !d_user_scale) opt.scaling = 1;       // (user factors can only come through set_scaling)
        V.cb = V.L + Sy.l_doubles;          // one pool: panels of in-place chain fronts live inside the cb part
        V.arena = nullptr; V.top_rhs = nullptr; V.rank = opt.rank; V.dbg = nullptr;
        if (knob_trace("clocks")) { if (!dalloc(&V.dbg, 128)) return false; }
        if (multi) { if (!dalloc(&V.arena, (size_t)arena_doubles) || !dalloc(&V.top_rhs, (size_t)toprhs_doubles)) return false; }
        V.pivtol = opt.pivtol; V.pivtol2 = std::max(opt.pivtol, opt.pivtolmax); V.small = opt.small; V.n = Sy.n; V.nnz_a = Sy.nnz_a; V.nsn = Sy.num_sn; V.xcd_affine = knob_disabled("xcd_affine") ? 0 : 1;
        // allow the large dynamic LDS sizes
        HIPCHK(hipFuncSetAttribute((const void*)k_big_diag_reg<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)(k_big_diag_reg<4, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)(k_front_reg<64, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)(k_front_reg<64, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)(k_front_reg<64, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)(k_front_reg<256, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)(k_front_reg<256, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)(k_front_reg<256, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)(k_front_reg<256, 6, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        // exact LDS need of the register-tiled front kernel per (level, class) bucket
        reg_lds.assign((size_t)Sy.num_levels * FC_COUNT, 0);
        for (int s = 0; s < Sy.num_sn; ++s) {
            const size_t m = Sy.sn_rowptr[s + 1] - Sy.sn_rowptr[s], k = Sy.sn_colptr[s + 1] - Sy.sn_colptr[s];
            const size_t ld = m | 1, ldi = k | 1;
            const size_t maxm = Sy.sn_class[s] == FC_WAVE ? 32 : (Sy.sn_class[s] == FC_LDS64 ? 64 : 128);
            const size_t need = (std::max(m * (m + 1) / 2, k * ld + k * ldi) + 4 * maxm + 3 * k) * sizeof(double) + 2 * k * sizeof(int) + 64;
            size_t& r = reg_lds[(size_t)Sy.sn_level[s] * FC_COUNT + Sy.sn_class[s]];
            r = std::max(r, need);
        }
        HIPCHK(hipFuncSetAttribute((const void*)k_big_trsm<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)k_big_trsm<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)k_big_diag_trsm, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)k_grp_fused, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)k_big_assemble2, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));      // (+ 1.3 KB of static LDS: the per-child tables)
        // per (level, BIG) bucket: largest front order / pivot count (launch geometry)
        big_maxm.assign(Sy.num_levels, 0); big_maxk.assign(Sy.num_levels, 0); big_tiles.assign(Sy.num_levels, 0); big_tiles64.assign(Sy.num_levels, 0);
        for (int s = 0; s < Sy.num_sn; ++s) if (Sy.sn_class[s] == FC_BIG) {
            const int lv = Sy.sn_level[s];
            big_maxm[lv] = std::max(big_maxm[lv], Sy.sn_rowptr[s + 1] - Sy.sn_rowptr[s]);
            big_maxk[lv] = std::max(big_maxk[lv], Sy.sn_colptr[s + 1] - Sy.sn_colptr[s]);
            big_tiles[lv] = std::max(big_tiles[lv], schur_tiles(Sy, s));
            big_tiles64[lv] = std::max(big_tiles64[lv], schur_tiles64(Sy, s));
        }
        // hipMemset runs on the legacy default stream; the solver's streams are non-blocking, i.e. NOT ordered behind it: without this the first
        // sweeps of a small system could meet flags / tagged messages left in recycled device memory by an earlier handle (or process) before
        // the zero fill has landed
        HIPCHK(hipDeviceSynchronize());
        lap("kernel attributes, zero fills landed");
        ready = true;
        if (keep && comm_kind == 2) make_subcomms();
        return true;
    }

    static int schur_tiles64(const Symbolic& Sy, int s) {
        const int mu = (Sy.sn_rowptr[s + 1] - Sy.sn_rowptr[s]) - (Sy.sn_colptr[s + 1] - Sy.sn_colptr[s]);
        const int nt = (mu + 63) / 64;
        return Sy.grp_rem[s] > 0 ? nt * ((Sy.grp_rem[s] + 63) / 64) : nt * (nt + 1) / 2;
    }
    static int tri_tiles(int nt) { const int t = nt * (nt + 1) / 2; return nt >= 12 ? (t + 7) / 8 * 8 : t; }    // large ones: multiple of 8 (XCD-aware order)
    // 128x128 tiles of the trailing update of front s: the whole lower triangle, or (not the last link of a chain group) only
    // the tile columns of the group's remaining panels
    static int schur_tiles(const Symbolic& Sy, int s) {
        const int mu = (Sy.sn_rowptr[s + 1] - Sy.sn_rowptr[s]) - (Sy.sn_colptr[s + 1] - Sy.sn_colptr[s]);
        const int nt = (mu + 127) / 128;
        return Sy.grp_rem[s] > 0 ? nt * ((Sy.grp_rem[s] + 127) / 128) : tri_tiles(nt);
    }
    static size_t trsm_lds(int kk) { return trsm_lds_bytes(kk, false); }
    static int schur_grid(int tiles) { return SCHUR_GM == 2 ? 16 * ((tiles + 7) / 8) : tiles; }      // workgroups of a k_big_schur launch over `tiles` 128 x 128 tiles (kernels_big.hip.inc: SCHUR_HALVES)
    int grid1d(long long n) const { long long g = (n + 255) / 256; return (int)std::min<long long>(std::max<long long>(g, 1), 2048); }


    // one (level, class) bucket of fronts
    bool launch_bucket(int lv, int fc, int b0, int b1, int top_mode, int mm, int kk, int tiles, int tiles64) {
        const int nb = b1 - b0;
        const size_t rl = reg_lds[(size_t)lv * FC_COUNT + fc];
        if (fc == FC_WAVE) {
            const bool sg = b0 == S->level_ptr[(size_t)lv * FC_COUNT + FC_WAVE];                            // single-GPU schedule: the bucket is sorted by order
            const int nt = sg ? tiny_split[lv] : 0;
            int fl = top_mode;
            if (V.fastpiv && wave_mmin[lv] <= 16) {      // fronts of order <= 16: four per wavefront on the static-order path first; what it accepts is skipped below
                const int n16 = sg ? tiny16[lv] : nb;
                // (the kernel raises qstat[4] for a front it rejects only when NO strict launch follows -- optimistic && n16 == nb: behind a strict
                // launch the rejected front is simply the strict kernel's, and raising the flag there made factor() repeat the whole factorisation
                // with the full schedule and drop the optimistic one for the life of the handle: ADVICE r04)
                if (n16 > 0) { LAUNCH(KK_FRONT_WAVE, k_front_dpp16, dim3((n16 + 3) / 4), dim3(64), 0, stream, V, b0, n16, top_mode, (optimistic && n16 == nb) ? 1 : 0); fl |= 2; }
                // OPTIMISTIC schedule: every front of the bucket has order <= 16 and the static-order kernel accepts (nearly) everything it is given
                // (LukVlE1 10^6: all 164 000 fronts) -- the strict launch behind it would find nothing to do, 4-8 us each, 21 of them per
                // factorisation.  It is left out; a front the kernel rejects raises qstat[4] and factor() runs the full schedule again.
                if (optimistic && n16 == nb) return true;
            }
            if (nt > 0) LAUNCH(KK_FRONT_WAVE, (k_front_reg<64, 2>), dim3(nt), dim3(64), rl, stream, V, b0, fl);
            if (nb - nt > 0) LAUNCH(KK_FRONT_WAVE, (k_front_reg<64, 4>), dim3(nb - nt), dim3(64), rl, stream, V, b0 + nt, fl);
        } else if (fc == FC_LDS64) {
            LAUNCH(KK_FRONT_LDS64, (k_front_reg<64, 8>), dim3(nb), dim3(64), rl, stream, V, b0, top_mode);
        } else if (fc == FC_LDS128) {
            const int nm = (b0 == S->level_ptr[(size_t)lv * FC_COUNT + FC_LDS128]) ? mid_split[lv] : 0;    // single-GPU schedule only
            const int fl = top_mode | (V.fastpiv ? 2 : 0);
            // (measured and not kept, r05: in the optimistic schedule the strict launch only on the fronts with > 16 pivots and on a device-built list of
            // the fronts the static-order launch rejected -- 342 of them on synth_1e6 -- instead of the whole bucket: 18.21 against 18.16 ms.  The
            // 60-150 us of the strict launches are those fronts' own latency chains, not the workgroups that find nothing to do.)
            if (V.fastpiv) {      // fronts with <= 16 pivots: static-order path first; what it accepts is skipped by the launch behind it
                // (<256, 6, true> cut to 128 VGPRs = 4 workgroups per CU: 2.24 -> 2.13 ms of front_lds128 on synth_1e6; the same cut of <256, 8, true> gained nothing and is gone)
                if (nm > 0) LAUNCH(KK_FRONT_LDS128, (k_front_reg<256, 6, true, 4>), dim3(nm), dim3(256), mid_lds[lv], stream, V, b0, top_mode);
                if (nb - nm > 0) LAUNCH(KK_FRONT_LDS128, (k_front_reg<256, 8, true>), dim3(nb - nm), dim3(256), rl, stream, V, b0 + nm, top_mode);
            }
            if (nm > 0) LAUNCH(KK_FRONT_LDS128, (k_front_reg<256, 6>), dim3(nm), dim3(256), mid_lds[lv], stream, V, b0, fl);
            if (nb - nm > 0) LAUNCH(KK_FRONT_LDS128, (k_front_reg<256, 8>), dim3(nb - nm), dim3(256), rl, stream, V, b0 + nm, fl);
        } else {
            const bool single = (b0 == S->level_ptr[(size_t)lv * FC_COUNT + FC_BIG]) && !multi;       // the single-GPU schedule
            if ((single || multi) && grouped && lv >= S->grp_cut_level) {
                // the chain groups whose first link sits on this level, one launch each (the multi-GPU schedules have their own group lists)
                if (!(single && lv_asm_skip[lv])) launch_assemble(mm, nb, b0, top_mode);
                return launch_groups(lv, single ? gs_single : (top_mode ? *gs_cur : gs_local));
            }
            if (!single) { const bool sm = mm <= 640; return launch_big(lv, b0, sm ? b1 : b0, b1, top_mode, mm, kk, sm ? tiles64 : 0, sm ? 0 : tiles, false); }
            return launch_big(lv, b0, b0 + big_split[lv], b1, top_mode, mm, kk, part_tiles[0][lv], part_tiles[1][lv], true);
        }
        return true;
    }
    // the chain groups whose first link sits on level lv: pivot blocks + leading blocks, rows below, rank-(<= 256) updates
    bool launch_groups(int lv, GrpSched& G) {
        const int b0 = G.g0[lv], b1 = G.g1[lv], bs = b0 + G.split[lv];
        if (b1 == b0) return true;
        {
            // as many 64-row blocks per row-block workgroup as it takes for the launch to fit the chip (one workgroup per CU)
            int rbw = 1;
            while (rbw < grp_rbw_max && (b1 - b0) * (4 + (G.nrb[lv] + rbw - 1) / rbw) > 256) rbw *= 2;
            const int nrbw = (G.nrb[lv] + rbw - 1) / rbw;
            const int st = ((b1 - b0) * (4 + nrbw) <= 256) ? 1 : 0;
            const size_t lds = std::max(diag_lds_bytes(64, 64), GRP_DB_BYTES + trsm_lds_bytes(64, st != 0));
            LAUNCH(KK_BIG_DIAG, k_grp_fused, dim3(b1 - b0, 4 + nrbw), dim3(256), lds, stream, V, b0, st, rbw, 0);
        }
        if (bs > b0 && G.tiles64[lv] > 0) LAUNCH(KK_BIG_SCHUR, k_big_schur64, dim3(G.tiles64[lv], bs - b0), dim3(256), 0, stream, V, b0, 0);
        if (b1 == bs) return true;
        const int nb = b1 - bs;
        if (la_pending) { HIPCHK(hipStreamWaitEvent(stream, la_last, 0)); la_pending = false; }      // a full update may touch what an earlier part 2 is still writing
        if (G.la2[lv] > 0) {      // (also while profile() records its events: the per-kernel times are those of the TIMED schedule, look-ahead stream included)
            // part 1 in 64 x 64 tiles where the 128 x 128 ones would leave most of the chip idle (k_big_schur_p1); the fronts of the list that are not split
            // get their whole update from a launch of their own then
            if (p1_small_tiles && G.la1[lv] * nb <= 512) {
                LAUNCH(KK_BIG_SCHUR, k_big_schur_p1, dim3(G.p1t[lv], nb), dim3(1024), 0, stream, V, bs);
                if (G.nsplit[lv] < nb) LAUNCH(KK_BIG_SCHUR, k_big_schur, dim3(schur_grid(G.la3[lv]), nb), dim3(SCHUR_NT), 0, stream, V, bs, 4, 0, 0);
            } else LAUNCH(KK_BIG_SCHUR, k_big_schur, dim3(schur_grid(G.la1[lv]), nb), dim3(SCHUR_NT), 0, stream, V, bs, 1, 0, 0);
            HIPCHK(hipEventRecord(G.evA[lv], stream));
            HIPCHK(hipStreamWaitEvent(stream2, G.evA[lv], 0));
            launch_part2(bs, nb, G.la2[lv]);
            HIPCHK(hipEventRecord(G.evB[lv], stream2));
            la_last = G.evB[lv]; la_pending = true;
        } else if (G.tiles[lv] > 0) LAUNCH(KK_BIG_SCHUR, k_big_schur, dim3(schur_grid(G.tiles[lv]), nb), dim3(SCHUR_NT), 0, stream, V, bs, 0, 0, 0);
        return true;
    }
    int grp_rbw_max = 8;
    int ntfuse = 0;
    // A handful of small fronts on a level whose bulk is elsewhere (synth_1e6: 3 fronts of order <= 128 next to 2 045 big ones; one front of order <= 64
    // next to 943 of order <= 128) is a launch of 1-3 workgroups running their strict pivot loops for 40-90 us with the chip idle.  The buckets of one level
    // are independent: in the eager (multi-stream) schedule such a bucket goes to the third stream, next to the level's main launches.
    std::vector<hipEvent_t> side_evF, side_evJ; bool side_on = !knob_disabled("side_small"); static constexpr int SIDE_MAX_FRONTS = 16;
    // runs of consecutive tree levels that hold nothing but one-wavefront fronts (order <= 32): one persistent data-flow launch each (k_front_df)
    struct DfRun { int lv0, lv1, tab0, nlev, nq; };
    std::vector<DfRun> df_runs; std::vector<int> df_run_at; const DfLevel* d_dftab = nullptr; int df_grid_cap = 0;
    bool df_on = !knob_disabled("front_df");
    int lc_levels = 0, lc_nchains = 0;      // leaf chains: the tree levels below lc_levels are lc_nchains chains of fronts of order <= 16 (k_leaf_chain)
    bool p1_small_tiles = !knob_disabled("p1_small");
    // part 2 of the split updates (second stream, next to the following group's pivot chain): 128 x 128 tiles on 16 wavefronts, or quarter tiles on 4
    // wavefronts that fit on the CUs a k_grp_fused workgroup occupies (kernels_big.hip.inc: k_big_schur_q / _w)
    void launch_part2(int b0, int nb, int ntiles) {
        const int nvirt = ((ntiles + 7) / 8) * 32;
        const int wgs = (int)std::min<long long>(nvirt, 4ll * la_wgs);
        LAUNCH_ON(KK_BIG_SCHUR, stream2, k_big_schur_q, dim3(wgs, nb), dim3(256), 0, stream2, V, b0, ntiles);
    }
    std::vector<char> asm_fast_ok;          // per launch-list entry: the front can take k_big_assemble2's fast path (the number of children it pulls; 0: it cannot)
    std::vector<int> h_asmcut;              // per launch-list entry: FrontMeta::asmcut
    // The column-chunk kernel pays where a level is MANY fronts of a few hundred rows (short columns: the one-wavefront-per-column kernel runs at the
    // latency of its load chain there) and every front can take its fast path (at most ASM_MAXCH children to pull, not an in-place link with other
    // children, no arena): measured 4.1 -> 2.0 ms per factorisation on synth_1e6, but 3.4 -> 8.7 ms on MBndryCntrl_3D 30, whose levels are a
    // handful of fronts of ~1000 rows with a dozen children each -- there the column kernel has four times the workgroups and no preamble.
    void launch_assemble(int mm, int nfronts, int b0, int top_mode) {
        const int ldi = (mm + 15) & ~15;                    // the children's inverse row maps of one front in LDS: ASM_MAXCH x ldi ints (78 KiB at the largest front of synth_1e6)
        // (... which has to FIT: a level of >= 32 fast-path fronts whose largest order exceeds ~6 780 -- a 3-D problem, or fronts enlarged by a
        // delayed-pivot edit -- would make the launch fail instead of falling back to the column kernel: ADVICE r04)
        // Round 5: a front whose contribution block is formed by its update (asmcut) has only its panel columns assembled; a launch of such fronts has no
        // workgroups behind the largest asmcut.  (MI355X_KKT_ASM2_WIDE: this kernel also for levels of a few dozen fronts with up to ASM_MAXCH = 16 children each
        // once only their panel columns are assembled -- the MBndryCntrl_3D family; measured slower there than the column kernel, 7.65 against 7.16 ms.)
        bool v2 = !top_mode;
        int maxch = 1, ncut = 0, cutmax = 0;
        for (int q = b0; q < b0 + nfronts && v2; ++q) { v2 = asm_fast_ok[q] != 0; maxch = std::max(maxch, (int)asm_fast_ok[q]); if (h_asmcut[q]) { ++ncut; cutmax = std::max(cutmax, h_asmcut[q] > 0 ? h_asmcut[q] : 0); } }
        v2 = v2 && (size_t)maxch * ldi * sizeof(int) <= (size_t)158 * 1024 && (nfronts >= 32 && maxch <= 6);
        if (!v2) { LAUNCH(KK_BIG_ASSEMBLE, k_big_assemble, dim3((mm + 3) / 4, nfronts), dim3(256), 0, stream, V, b0, top_mode); return; }
        const int ncols = (ncut == nfronts) ? std::min(mm, cutmax) : mm;      // (every front stops at its asmcut: no workgroups for the columns behind the largest of them)
        if (ncols <= 0) return;                                               // (every front of the list has its whole block formed by its update: nothing to assemble -- a grid of 0 workgroups is refused by the runtime, met on the multi-rank schedule)
        LAUNCH(KK_BIG_ASSEMBLE, k_big_assemble2, dim3((ncols + ASM_CH - 1) / ASM_CH, nfronts), dim3(256), (size_t)maxch * ldi * sizeof(int), stream, V, b0, top_mode, ldi, maxch);
    }
    // the big-front launches of one level: assembly, pivot blocks and TRSM over the whole list [b0, b1); the trailing update with
    // 64 x 64 tiles / 256 threads on the fronts [b0, bs) of order <= 1024 (a handful of tiles, K = 16..64 each) and with
    // 128 x 128 tiles / 1024 threads on [bs, b1)
    bool launch_big(int lv, int b0, int bs, int b1, int top_mode, int mm, int kk, int tiles_small, int tiles, bool single) {
        const int nball = b1 - b0;
        if (!(single && lv_asm_skip[lv])) launch_assemble(mm, nball, b0, top_mode);
        const int nrb = (mm + 63) / 64;
        if (fuse_dt && (single || multi) && kk <= 64 && nball * (1 + nrb) <= fuse_dt_maxwg) {      // (multi-GPU: local subtrees and replicated top alike; the per-kernel profile books the fused launch under the pivot blocks)
            // few fronts on the level: pivot block + panel solve in one flag-synchronised launch (k_big_diag_trsm)
            const size_t lds = std::max(diag_lds_bytes(kk, 64),
                                        trsm_lds_bytes(kk, true));
            const int ntu = (lv_narrow_tiles[lv] > 0 && nball * (1 + nrb + lv_narrow_tiles[lv]) <= 256) ? lv_narrow_tiles[lv] : 0;      // (one workgroup per CU: the far part of a group-end update may own the rest of the chip)
            LAUNCH(KK_BIG_DIAG, k_big_diag_trsm, dim3(nball, 1 + nrb + ntu), dim3(256), lds, stream, V, b0, nrb);
            if (ntu > 0) return true;         // ... and so did the narrow updates
            goto updates;
        }
        if (kk <= 64) LAUNCH(KK_BIG_DIAG, k_big_diag_reg<4>, dim3(nball), dim3(256), diag_lds_bytes(kk, 64), stream, V, b0);
        else          LAUNCH(KK_BIG_DIAG, (k_big_diag_reg<4, 1024>), dim3(nball), dim3(1024), diag_lds_bytes(kk, 128), stream, V, b0);
        if (kk <= 64) LAUNCH(KK_BIG_TRSM, k_big_trsm<false>, dim3((mm + 63) / 64, nball), dim3(256), trsm_lds(kk), stream, V, b0, 0);
        else          LAUNCH(KK_BIG_TRSM, k_big_trsm<true>, dim3((mm + 63) / 64, nball), dim3(256), trsm_lds(kk), stream, V, b0, 0);
      updates:
        if (bs > b0 && tiles_small > 0) LAUNCH(KK_BIG_SCHUR, k_big_schur64, dim3(tiles_small, bs - b0), dim3(256), 0, stream, V, b0, 0);
        if (b1 == bs) return true;
        const int nb = b1 - bs;
        b0 = bs;
        if (single && la_full[lv] && la_pending) {       // a full update may touch what an earlier part 2 is still writing
            HIPCHK(hipStreamWaitEvent(stream, la_last, 0)); la_pending = false;
        }
        if (single && la_tiles2[lv] > 0) {
            LAUNCH(KK_BIG_SCHUR, k_big_schur, dim3(schur_grid(la_tiles1[lv]), nb), dim3(SCHUR_NT), 0, stream, V, b0, 1, 0, 0);
            HIPCHK(hipEventRecord(la_evA[lv], stream));
            HIPCHK(hipStreamWaitEvent(stream2, la_evA[lv], 0));
            launch_part2(b0, nb, la_tiles2[lv]);
            HIPCHK(hipEventRecord(la_evB[lv], stream2));
            la_last = la_evB[lv]; la_pending = true;
        } else if (tiles > 0) LAUNCH(KK_BIG_SCHUR, k_big_schur, dim3(schur_grid(tiles), nb), dim3(SCHUR_NT), 0, stream, V, b0, 0, 0, 0);
        return true;
    }

    // symmetric scaling of the gathered values (mode 0 none / 1 Ruiz, 4 Jacobi-style sweeps ping-ponging between two buffers /
    // 2 the caller's factors) and the column norms of the scaled matrix that anchor the zero-pivot test
    void enqueue_scaling() {
        const Symbolic& Sy = *S; const int n = Sy.n;
        if (keep_scale_now) {      // the factors (and the column-norm scale, ~1 by construction) of the last equilibration
            LAUNCH(KK_GATHER_SCALE, k_apply_scale, dim3(grid1d(Sy.nnz_a)), dim3(256), 0, stream, V);
            return;
        }
        if (opt.scaling == 1) scale_valid = true;
        // Ruiz on short rows (LukVl: 6 entries per row): the row view is written by the first sweep (-16 us of 350; at 39 entries per row the flat pass + sweep are faster: +37 us)
        const bool fuse0 = opt.scaling == 1 && (long long)V.rslot_len < 16ll * n;
        if (!fuse0) LAUNCH(KK_GATHER_SCALE, k_abs_rowview, dim3(grid1d(V.rslot_len)), dim3(256), 0, stream, V);
        if (opt.scaling >= 2) LAUNCH(KK_GATHER_SCALE, k_user_scale, dim3(grid1d(n)), dim3(256), 0, stream, V, (const double*)d_user_scale);     // 2: the caller's factors, 3: matching (computed just before)
        else if (opt.scaling) {
            const int lpr = (long long)V.rslot_len < 8ll * n ? 2 : 8;      // short rows (LukVl: ~5 entries): 2 lanes per row (measured 0.89 / 0.82 / 0.80 ms per factorisation at 8 / 4 / 2)
            auto sweeps = [&](auto tag) {
                constexpr int L = decltype(tag)::value;
                if (fuse0) LAUNCH(KK_GATHER_SCALE, k_abs_rowview_sweep0<L>, dim3(grid1d((long long)L * n)), dim3(256), 0, stream, V, V.scale2);
                else LAUNCH(KK_GATHER_SCALE, k_ruiz_sweep<L>, dim3(grid1d((long long)L * n)), dim3(256), 0, stream, V, (const double*)nullptr, V.scale2, (double*)nullptr);
                LAUNCH(KK_GATHER_SCALE, k_ruiz_sweep<L>, dim3(grid1d((long long)L * n)), dim3(256), 0, stream, V, (const double*)V.scale2, V.scale, (double*)nullptr);
                LAUNCH(KK_GATHER_SCALE, k_ruiz_sweep<L>, dim3(grid1d((long long)L * n)), dim3(256), 0, stream, V, (const double*)V.scale, V.scale2, (double*)nullptr);
                LAUNCH(KK_GATHER_SCALE, k_ruiz_sweep<L>, dim3(grid1d((long long)L * n)), dim3(256), 0, stream, V, (const double*)V.scale2, V.scale, V.cnorm);
            };
            if (lpr == 2) sweeps(std::integral_constant<int, 2>()); else sweeps(std::integral_constant<int, 8>());
        } else LAUNCH(KK_GATHER_SCALE, k_fill, dim3(grid1d(n)), dim3(256), 0, stream, V.scale, 1.0, (long long)n);
        if (opt.scaling != 1) LAUNCH(KK_GATHER_SCALE, k_colnorm, dim3(grid1d(8ll * n)), dim3(256), 0, stream, V, opt.scaling ? 1 : 0);      // (Ruiz: ~1 by construction, written by the last sweep)
        if (opt.scaling) LAUNCH(KK_GATHER_SCALE, k_apply_scale, dim3(grid1d(Sy.nnz_a)), dim3(256), 0, stream, V);
    }
    bool enqueue_factor() {
        const Symbolic& Sy = *S;
        const int n = Sy.n;
        LAUNCH(KK_STATS, k_factor_prologue, dim3(grid1d(n)), dim3(256), 0, stream, V);
        LAUNCH(KK_GATHER_SCALE, k_gather_values, dim3(grid1d(Sy.nnz_a)), dim3(256), 0, stream, V);
        enqueue_scaling();
        const int lc = (optimistic && lc_levels > 0) ? lc_levels : 0;      // the leaf chains: one launch for their levels (optimistic schedule only: no strict kernel behind it)
        if (lc > 0) LAUNCH(KK_FRONT_WAVE, k_leaf_chain, dim3((lc_nchains + 3) / 4), dim3(64), 0, stream, V, lc_nchains);
        for (int lv = lc; lv < Sy.num_levels; ++lv) {
            if (optimistic && df_run_at[lv] >= 0) {      // a run of levels of one-wavefront fronts: one persistent data-flow launch (optimistic schedule only: no strict kernels behind it)
                const DfRun& R = df_runs[df_run_at[lv]];
                LAUNCH(KK_FRONT_WAVE, k_front_df, dim3(std::min(R.nq, df_grid_cap)), dim3(64), 0, stream, V, d_dftab + R.tab0, R.nlev, R.nq);
                lv = R.lv1; continue;
            }
            if (Sy.cb_window > 0 && lv % Sy.cb_window == 0) {      // recycled contribution blocks (symbolic.cpp step 12a): whatever earlier levels put on the look-ahead streams is
                if (la_pending) { HIPCHK(hipStreamWaitEvent(stream, la_last, 0)); la_pending = false; }      // through before a level >= lv may write into the space of a block they read
            }
            int lvl_fronts = 0, lvl_max = 0;
            for (int fc = 0; fc < FC_COUNT; ++fc) { const int nb = Sy.level_ptr[(size_t)lv * FC_COUNT + fc + 1] - Sy.level_ptr[(size_t)lv * FC_COUNT + fc]; lvl_fronts += nb; lvl_max = std::max(lvl_max, nb); }
            bool side_open = false;
            for (int fc = 0; fc < FC_COUNT; ++fc) {
                const int b0 = Sy.level_ptr[(size_t)lv * FC_COUNT + fc], b1 = Sy.level_ptr[(size_t)lv * FC_COUNT + fc + 1];
                if (b1 == b0) continue;
                // (the eager schedule only: a multi-stream hipGraph is what factor_once() avoids; not next to the three-stream chain look-ahead)
                const bool side = side_on && la_any && !multi && fc != FC_BIG && b1 - b0 <= SIDE_MAX_FRONTS && lvl_max >= 8 * (b1 - b0) && lvl_fronts > b1 - b0;
                if (!side) { launch_bucket(lv, fc, b0, b1, 0, big_maxm[lv], big_maxk[lv], big_tiles[lv], big_tiles64[lv]); continue; }
                if ((int)side_evF.size() < Sy.num_levels) { side_evF.resize(Sy.num_levels, nullptr); side_evJ.resize(Sy.num_levels, nullptr); }
                if (!side_evF[lv]) { HIPCHK(hipEventCreateWithFlags(&side_evF[lv], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&side_evJ[lv], hipEventDisableTiming)); }
                if (!side_open) { HIPCHK(hipEventRecord(side_evF[lv], stream)); HIPCHK(hipStreamWaitEvent(stream3, side_evF[lv], 0)); side_open = true; }      // behind the level below
                std::swap(stream, stream3);
                const bool ok = launch_bucket(lv, fc, b0, b1, 0, big_maxm[lv], big_maxk[lv], big_tiles[lv], big_tiles64[lv]);
                std::swap(stream, stream3);
                if (!ok) return false;
            }
            if (side_open) { HIPCHK(hipEventRecord(side_evJ[lv], stream3)); HIPCHK(hipStreamWaitEvent(stream, side_evJ[lv], 0)); }      // the level above reads what they wrote
        }
        if (la_pending) { HIPCHK(hipStreamWaitEvent(stream, la_last, 0)); la_pending = false; }
        LAUNCH(KK_STATS, k_zero_i32, dim3(1), dim3(64), 0, stream, d_stats, 4);
        LAUNCH(KK_STATS, k_reduce_stats, dim3(std::min(64, (Sy.num_sn + 255) / 256)), dim3(256), 0, stream, V.fstat, V.apfail, V.sn_owner, Sy.num_sn, -2, d_stats);
        HIPCHK(hipGetLastError());
        return true;
    }

    bool norestore_on = !knob_disabled("norestore");      // (development knob: keep the safety copies of the pivot blocks in the optimistic schedule too)
    bool optimistic = false;                 // (set per factorisation: see launch_bucket)
    bool optimistic_ok = true;               // the optimistic schedule may be tried (false while a back-off runs: see factor())
    int  opt_backoff = 0, opt_wait = 0;      // after a fall-back: opt_wait factorisations on the full schedule, then one more optimistic try; every repeat doubles the wait (8 .. 256)
    hipGraphExec_t g_factor_full = nullptr;  // the schedule with every strict launch (g_factor: the optimistic one)
    bool factor(const double* dvals, bool reuse, FactorStats& st) {
        DeviceGuard guard(dev);
        if (!ready) { if (err_.empty()) err_ = "factor: solver not set up (no device?)"; return false; }
        if (multi) return factor_dist(dvals, reuse, st);          // needs a communicator (set_comm_*), fails loudly otherwise
        static const bool opt_off = knob_disabled("optimistic");
        if (!optimistic_ok && opt_wait > 0 && --opt_wait == 0) optimistic_ok = true;   // (one rejection early in an Ipopt run must not cost the optimistic schedule for the life of the structure)
        optimistic = V.fastpiv && !opt_off && !prof_on && optimistic_ok;
        if (!factor_once(dvals, reuse, st)) return false;
        if (optimistic && (h_stats[8] != 0 || h_stats[9] != 0)) { // some front was left for a strict launch that was not there / a pivot block without a safety copy was rejected: the full schedule, same values
            optimistic = false; optimistic_ok = false;            // ... and for the next opt_wait factorisations: a matrix family that needs the strict kernels once tends to need them again
            opt_backoff = opt_backoff == 0 ? 8 : std::min(256, 2 * opt_backoff); opt_wait = opt_backoff;
            if (opt.verbose) fprintf(stderr, h_stats[10] ? "[mi355x_kkt] factor: a wait of the data-flow launch over the small-front levels timed out (device shared?), running the full schedule\n"
                                                           : "[mi355x_kkt] factor: the optimistic schedule met a front for the strict kernels, running the full one\n");
            return factor_once(nullptr, true, st);
        }
        return true;
    }
    bool factor_once(const double* dvals, bool reuse, FactorStats& st) {
        const Symbolic& Sy = *S;
        V.pivtol = opt.pivtol; V.pivtol2 = std::max(opt.pivtol, opt.pivtolmax); V.small = opt.small;
        V.norestore = (optimistic && norestore_on) ? 1 : 0;
        if (!reuse) {
            if (dvals) HIPCHK(hipMemcpyAsync((void*)V.tvals, dvals, Sy.nnz_in * sizeof(double), hipMemcpyDeviceToDevice, stream));
            else       HIPCHK(hipMemcpyAsync((void*)V.tvals, h_vals, Sy.nnz_in * sizeof(double), hipMemcpyHostToDevice, stream));
            have_values = true;
        } else if (!have_values) { err_ = "refactor: no values on the device yet"; return false; }
        if (want_matching() && Sy.n > 0) { if (!run_matching()) return false; match_valid = true; }
        HIPCHK(hipEventRecord(ev0, stream));
        // A factorisation with look-ahead forks onto the second stream: it is launched eagerly (measured equal to the graph
        // replay on these ~10^3-launch sequences, whose kernels are long), because a two-stream hipGraph replays up to 1.5x
        // slower once another solver's graphs have been created and destroyed in the same process (ROCm 7.2).
        if (opt.use_graph && !la_any) {
            if (graph_pivtol != V.pivtol || graph_pivtol2 != V.pivtol2) {      // (both schedules were captured with the old thresholds)
                destroy_factor_graphs();
            }
            hipGraphExec_t& g_factor = keep_scale_now ? (optimistic ? g_factor_keep : g_factor_full_keep) : (optimistic ? this->g_factor : g_factor_full);
            if (!g_factor) {
                hipGraph_t g = nullptr;
                HIPCHK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
                bool ok = enqueue_factor();
                hipError_t e = hipStreamEndCapture(stream, &g);
                if (!ok) return false;
                if (e != hipSuccess) { err_ = std::string("hipStreamEndCapture: ") + hipGetErrorString(e); return false; }
                HIPCHK(hipGraphInstantiate(&g_factor, g, nullptr, nullptr, 0));
                (void)hipGraphDestroy(g);
                graph_pivtol = V.pivtol; graph_pivtol2 = V.pivtol2;
                // the events recorded before capture are still valid; re-record for timing accuracy
                HIPCHK(hipEventRecord(ev0, stream));
            }
            HIPCHK(hipGraphLaunch(g_factor, stream));
        } else {
            if (!enqueue_factor()) return false;
        }
        HIPCHK(hipEventRecord(ev1, stream));
        HIPCHK(hipMemcpyAsync(h_stats, d_stats, 12 * sizeof(int), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ev0, ev1)); factor_ms = ms;
        st.num_neg = h_stats[0]; st.num_zero = h_stats[1]; st.num_two = h_stats[2]; st.num_small = h_stats[3]; st.u_sensitive = h_stats[4]; st.num_fast = h_stats[7];
        if (h_stats[5] != 0) { err_ = "factor: a panel workgroup timed out waiting for its pivot block"; return false; }
        return true;
    }
    double graph_pivtol = -1.0, graph_pivtol2 = -1.0;

    // The sweeps only touch the solver's own buffers, so ONE captured graph serves every right-hand side: the two kernels
    // that see the caller's pointers (k_load_rhs / k_store_sol) are launched eagerly around the replay.
    bool enqueue_solve(const double* dsrc, double* drhs) {
        LAUNCH(KK_SOLVE_PERM, k_load_rhs, dim3(grid1d(S->n)), dim3(256), 0, stream, V, dsrc);
        if (!enqueue_solve_core()) return false;
        LAUNCH(KK_SOLVE_PERM, k_store_sol, dim3(grid1d(S->n)), dim3(256), 0, stream, V, drhs);
        HIPCHK(hipGetLastError());
        return true;
    }
    bool enqueue_solve_core() {
        const Symbolic& Sy = *S;
        const int n = Sy.n;
        const int nref = opt.refine_steps > 0 ? opt.refine_steps : 0;
        if (nref > 0) LAUNCH(KK_SOLVE_PERM, k_save_rhs, dim3(grid1d(n)), dim3(256), 0, stream, V);
        for (int pass = 0; pass <= nref; ++pass) {
            if (pass > 0) {      // xacc (+)= xw ; xw = bw - K xacc ; solve again for the correction
                LAUNCH(KK_SOLVE_PERM, k_refine_residual, dim3(grid1d(n)), dim3(256), 0, stream, V, pass == 1 ? 1 : 0);
                LAUNCH(KK_SOLVE_PERM, k_refine_spmv, dim3(grid1d(n)), dim3(256), 0, stream, V);
            }
            auto lds_solve = [](int mmax, int kmax) { return (size_t)(mmax + 3 * kmax) * sizeof(double) + 16; };
            if (!chain_segs.empty()) LAUNCH(KK_SOLVE_PERM, k_bump_epoch, dim3(1), dim3(64), 0, stream, V.sepoch);
            int lcs = (pair_solve && lc_levels > 0) ? lc_levels : 0;
            for (int lv = 0; lv < lcs; ++lv) if (seg_at_lv0[lv] >= 0) { lcs = 0; break; }      // (a small tree: one data-flow sweep already covers these levels)
            if (lcs > 0) LAUNCH(KK_FWD_WAVE, k_fwd_leafchain, dim3((lc_nchains + 3) / 4), dim3(64), 0, stream, V, lc_nchains);
            for (int lv = lcs; lv < Sy.num_levels; ++lv) {
                if (seg_at_lv0[lv] >= 0) {      // a run of pure chain levels: one sync-free launch for all of them
                    const ChainSeg& sg = chain_segs[seg_at_lv0[lv]];
                    if (gate_on && !gate_entered) { if (gate_valid) HIPCHK(hipStreamWaitEvent(stream, gate_last, 0)); gate_entered = true; }      // (solve contexts: one chain section at a time)
                    LAUNCH(KK_FWD_BIG, k_fwd_chain, dim3(sg.nwg_f), dim3(320), 0, stream, V, sg.wgf0);
                    lv = sg.lv1; continue;
                }
                for (int fc = 0; fc < FC_COUNT; ++fc) {
                    const int b0 = Sy.level_ptr[(size_t)lv * FC_COUNT + fc], b1 = Sy.level_ptr[(size_t)lv * FC_COUNT + fc + 1];
                    if (b1 == b0) continue;
                    if (fc == FC_WAVE && pair_solve && wave_mmax[lv] <= 16) LAUNCH(KK_FWD_WAVE, (k_fwd_pair<16, 16>), dim3((b1 - b0 + 3) / 4), dim3(64), 0, stream, V, b0, b1 - b0);
                    else if (fc == FC_WAVE && pair_solve && wave_kmax[lv] <= 16) LAUNCH(KK_FWD_WAVE, (k_fwd_pair<16>), dim3((b1 - b0 + 1) / 2), dim3(64), 0, stream, V, b0, b1 - b0);
                    else if (fc == FC_WAVE && pair_solve)                   LAUNCH(KK_FWD_WAVE, (k_fwd_pair<32>), dim3((b1 - b0 + 1) / 2), dim3(64), 0, stream, V, b0, b1 - b0);
                    else if (fc == FC_WAVE)   LAUNCH(KK_FWD_WAVE, (k_fwd<64>),  dim3(b1 - b0), dim3(64),  lds_solve(32, 32),   stream, V, b0, 0);
                    else if (fc == FC_LDS64)  LAUNCH(KK_FWD_LDS,  (k_fwd<64>),  dim3(b1 - b0), dim3(64),  lds_solve(64, 64),   stream, V, b0, 0);
                    else if (fc == FC_LDS128) LAUNCH(KK_FWD_LDS,  (k_fwd<256>), dim3(b1 - b0), dim3(256), lds_solve(128, 128), stream, V, b0, 0);
                    else if (big_last1[lv] > big_last0[lv]) {
                           const int g0 = big_last0[lv], ng = big_last1[lv] - g0;
                           if (lv_allsolo[lv] && big_maxk[lv] <= 64) LAUNCH(KK_FWD_BIG, k_fwd_solo64, dim3((big_maxm[lv] + 63) / 64 + 1, ng), dim3(256), 0, stream, V, g0);
                           else if (lv_allsolo[lv]) LAUNCH(KK_FWD_BIG, k_fwd_solo, dim3((big_maxm[lv] + 63) / 64 + 1, ng), dim3(256), 0, stream, V, g0);
                           else { LAUNCH(KK_FWD_BIG, k_fwd_grp, dim3(ng), dim3(256), 0, stream, V, g0, 0);
                                  LAUNCH(KK_FWD_BIG_UPD, k_fwd_grp_upd, dim3((big_maxm[lv] + 63) / 64, ng), dim3(256), 0, stream, V, g0); } }
                }
            }
            for (int lv = Sy.num_levels - 1; lv >= lcs; --lv) {
                if (seg_at_lv1[lv] >= 0) {
                    const ChainSeg& sg = chain_segs[seg_at_lv1[lv]];
                    LAUNCH(KK_BWD_BIG, k_bwd_chain, dim3(sg.nwg_b), dim3(320), 0, stream, V, sg.wgb0);
                    if (gate_on && gate_mine) { HIPCHK(hipEventRecord(gate_mine, stream)); gate_last = gate_mine; gate_valid = true; }
                    lv = sg.lv0; continue;
                }
                for (int fc = 0; fc < FC_COUNT; ++fc) {
                    const int b0 = Sy.level_ptr[(size_t)lv * FC_COUNT + fc], b1 = Sy.level_ptr[(size_t)lv * FC_COUNT + fc + 1];
                    if (b1 == b0) continue;
                    if (fc == FC_WAVE && pair_solve && wave_mmax[lv] <= 16) LAUNCH(KK_BWD_WAVE, (k_bwd_pair<16, 16>), dim3((b1 - b0 + 3) / 4), dim3(64), 0, stream, V, b0, b1 - b0);
                    else if (fc == FC_WAVE && pair_solve && wave_kmax[lv] <= 16) LAUNCH(KK_BWD_WAVE, (k_bwd_pair<16>), dim3((b1 - b0 + 1) / 2), dim3(64), 0, stream, V, b0, b1 - b0);
                    else if (fc == FC_WAVE && pair_solve)                   LAUNCH(KK_BWD_WAVE, (k_bwd_pair<32>), dim3((b1 - b0 + 1) / 2), dim3(64), 0, stream, V, b0, b1 - b0);
                    else if (fc == FC_WAVE)   LAUNCH(KK_BWD_WAVE, (k_bwd<64>),  dim3(b1 - b0), dim3(64),  lds_solve(32, 32),   stream, V, b0);
                    else if (fc == FC_LDS64)  LAUNCH(KK_BWD_LDS,  (k_bwd<64>),  dim3(b1 - b0), dim3(64),  lds_solve(64, 64),   stream, V, b0);
                    else if (fc == FC_LDS128) LAUNCH(KK_BWD_LDS,  (k_bwd<256>), dim3(b1 - b0), dim3(256), lds_solve(128, 128), stream, V, b0);
                    else if (big_last1[lv] > big_last0[lv]) {
                           const int g0 = big_last0[lv], ng = big_last1[lv] - g0;
                           if (!Sy.solve_group && big_maxk[lv] <= 64) {      // one link per unit: load-hoisted kernels
                               LAUNCH(KK_BWD_BIG_DOT, k_bwd_dot64, dim3((big_maxm[lv] + 255) / 256, ng), dim3(256), 0, stream, V, g0);
                               LAUNCH(KK_BWD_BIG, k_bwd_fin64, dim3(ng), dim3(256), 0, stream, V, g0);
                           } else {
                               LAUNCH(KK_BWD_BIG_DOT, k_bwd_grp_dot, dim3((big_maxm[lv] + 255) / 256, ng), dim3(256), 0, stream, V, g0);
                               LAUNCH(KK_BWD_BIG, k_bwd_grp, dim3(ng), dim3(256), 0, stream, V, g0); } }
                }
            }
            if (lcs > 0) LAUNCH(KK_BWD_WAVE, k_bwd_leafchain, dim3((lc_nchains + 3) / 4), dim3(64), 0, stream, V, lc_nchains);
        }
        if (nref > 0) LAUNCH(KK_SOLVE_PERM, k_refine_finish, dim3(grid1d(n)), dim3(256), 0, stream, V);
        HIPCHK(hipGetLastError());
        return true;
    }

    // ---- several right-hand sides (MultiSolve with nrhs > 1, IpSparseSymLinearSolverInterface.hpp:190; the reference hands nrhs straight to ONE ma97_solve,
    //      IpMa97SolverInterface.cpp:790,805): a single solve is a LATENCY chain (1.8 of the ~6 TB/s a stream of L could take), so the right-hand sides are not
    //      solved one behind the other: up to SOLVE_CTX of them are in flight at once, each in a CONTEXT of its own -- its own stream, work vectors, message
    //      tags, flags and epoch -- through the same kernels and the same launch sequence (the same bits per column as a solve on its own).  The panels of L that
    //      one column's sweep has just pulled through L2 / the Infinity Cache are what the neighbouring columns' sweeps read next.  Context 0 is the solver's own
    //      workspace and stream; the others are allocated at the first call that needs them (a caller with one right-hand side never pays for them). ----
    static constexpr int SOLVE_CTX = 3;      // (streams of ONE priority share a hardware queue and are serialised by it: one context per priority level the runtime offers)
    struct SolveCtx { hipStream_t strm = nullptr; hipEvent_t done = nullptr; hipGraphExec_t graph = nullptr;
                      double *xw = nullptr, *cvec = nullptr, *bw = nullptr, *xacc = nullptr, *gpart = nullptr, *zb = nullptr, *ybuf = nullptr, *dpart = nullptr;
                      int *sflag_t = nullptr, *sflag_dot = nullptr, *sflag_b = nullptr, *sepoch = nullptr; v2d *ytag = nullptr, *xtag = nullptr; };
    SolveCtx sctx[SOLVE_CTX]; int nsctx = 1; hipEvent_t sctx_fork = nullptr;
    int sctx_verdict = 0; bool sctx_tuning = false;      // 0: not measured for this structure yet, 1: contexts, 2: one after the other
    // The data-flow sweeps over the top of the tree (k_fwd_chain / k_bwd_chain) are PERSISTENT launches whose workgroups wait for one another; the argument that
    // a waiting workgroup never holds up the one it waits for (lower index = dispatched first) is an argument about ONE such launch on the device.  Two of them
    // on two hardware queues did starve each other (measured, round 6: time-outs of the bounded waits as soon as the solver's own high-priority stream and a
    // context's stream each had a sweep in flight; three contexts on streams of one priority share a hardware queue and were serialised by it).  So the
    // contexts take turns through a GATE for their chain section -- from the first forward chain launch to the last backward one --, everything below it (the
    // level launches of the bulk of the tree, which hold 2/3 of L) overlaps freely.  Needs eager launches: the gate is an event between streams.
    bool gate_on = false, gate_valid = false, gate_entered = false; hipEvent_t gate_last = nullptr, gate_mine = nullptr; hipEvent_t gate_ev[SOLVE_CTX] = {nullptr, nullptr, nullptr};
    size_t n_sflag_dot = 1, n_dpart = 1, n_sflag_t = 1, n_sflag_b = 1, n_tag = 1;      // element counts of the flag / tag arrays (set where they are allocated)
    void sctx_release() {      // (the buffers are in `allocs`: release() frees them)
        for (int c = 0; c < SOLVE_CTX; ++c) if (gate_ev[c]) { (void)hipEventDestroy(gate_ev[c]); gate_ev[c] = nullptr; }
        gate_on = gate_valid = gate_entered = false; gate_last = gate_mine = nullptr; sctx_verdict = 0;
        for (int c = 1; c < SOLVE_CTX; ++c) { if (sctx[c].graph) (void)hipGraphExecDestroy(sctx[c].graph); if (sctx[c].done) (void)hipEventDestroy(sctx[c].done);
                                               if (sctx[c].strm) (void)hipStreamDestroy(sctx[c].strm); sctx[c] = SolveCtx(); }
        if (sctx_fork) { (void)hipEventDestroy(sctx_fork); sctx_fork = nullptr; }
        nsctx = 1;
    }
    bool sctx_ensure(int want) {
        want = std::min(want, SOLVE_CTX);
        const Symbolic& Sy = *S;
        if (!sctx_fork) HIPCHK(hipEventCreateWithFlags(&sctx_fork, hipEventDisableTiming));
        for (int c = 0; c < want; ++c) if (!gate_ev[c]) HIPCHK(hipEventCreateWithFlags(&gate_ev[c], hipEventDisableTiming));
        for (int c = nsctx; c < want; ++c) {
            SolveCtx& X = sctx[c];
            { int plo = 0, phi = 0; (void)hipDeviceGetStreamPriorityRange(&plo, &phi);      // (numerically lower = higher priority; the solver's own stream has phi)
              HIPCHK(hipStreamCreateWithPriority(&X.strm, hipStreamNonBlocking, c == 1 ? (plo + phi) / 2 : plo)); }
            HIPCHK(hipEventCreateWithFlags(&X.done, hipEventDisableTiming));
            if (!dalloc(&X.xw, Sy.n) || !dalloc(&X.ybuf, Sy.n) || !dalloc(&X.zb, Sy.n) || !dalloc(&X.bw, Sy.n) || !dalloc(&X.xacc, Sy.n) || !dalloc(&X.cvec, (size_t)Sy.cvec_doubles) ||
                !dalloc(&X.gpart, (size_t)Sy.gpart_doubles) || !dalloc(&X.sflag_dot, n_sflag_dot) || !dalloc(&X.dpart, n_dpart) || !dalloc(&X.sflag_t, n_sflag_t) || !dalloc(&X.sflag_b, n_sflag_b) ||
                !dalloc(&X.ytag, n_tag) || !dalloc(&X.xtag, n_tag) || !dalloc(&X.sepoch, 4)) return false;
            HIPCHK(hipDeviceSynchronize());      // (dalloc's zero fills ran on the default stream)
            nsctx = c + 1;
        }
        return true;
    }
    // the solver works in context c from here on (c = 0: back to its own workspace and stream); `saved` keeps what context 0 owns
    struct SctxSaved { hipStream_t strm; hipGraphExec_t graph; SolveCtx w; };
    void sctx_enter(int c, SctxSaved& sv) {
        sv.strm = stream; sv.graph = g_solve;
        sv.w.xw = V.xw; sv.w.cvec = V.cvec; sv.w.bw = V.bw; sv.w.xacc = V.xacc; sv.w.gpart = V.gpart; sv.w.zb = V.zb; sv.w.ybuf = V.ybuf; sv.w.dpart = V.dpart;
        sv.w.sflag_t = V.sflag_t; sv.w.sflag_dot = V.sflag_dot; sv.w.sflag_b = V.sflag_b; sv.w.sepoch = V.sepoch; sv.w.ytag = V.ytag; sv.w.xtag = V.xtag;
        const SolveCtx& X = sctx[c];
        stream = X.strm; g_solve = X.graph;
        V.xw = X.xw; V.cvec = X.cvec; V.bw = X.bw; V.xacc = X.xacc; V.gpart = X.gpart; V.zb = X.zb; V.ybuf = X.ybuf; V.dpart = X.dpart;
        V.sflag_t = X.sflag_t; V.sflag_dot = X.sflag_dot; V.sflag_b = X.sflag_b; V.sepoch = X.sepoch; V.ytag = X.ytag; V.xtag = X.xtag;
    }
    void sctx_leave(int c, const SctxSaved& sv) {
        sctx[c].graph = g_solve;          // (captured on first use)
        stream = sv.strm; g_solve = sv.graph;
        V.xw = sv.w.xw; V.cvec = sv.w.cvec; V.bw = sv.w.bw; V.xacc = sv.w.xacc; V.gpart = sv.w.gpart; V.zb = sv.w.zb; V.ybuf = sv.w.ybuf; V.dpart = sv.w.dpart;
        V.sflag_t = sv.w.sflag_t; V.sflag_dot = sv.w.sflag_dot; V.sflag_b = sv.w.sflag_b; V.sepoch = sv.w.sepoch; V.ytag = sv.w.ytag; V.xtag = sv.w.xtag;
    }
    bool solve_one(const double* src, double* col) {      // one right-hand side in the CURRENT context (stream, V's work vectors, g_solve)
        if (opt.use_graph) {
            if (!g_solve) {
                hipGraph_t g = nullptr;
                HIPCHK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
                bool ok = enqueue_solve_core();
                hipError_t e = hipStreamEndCapture(stream, &g);
                if (!ok) return false;
                if (e != hipSuccess) { err_ = std::string("hipStreamEndCapture: ") + hipGetErrorString(e); return false; }
                HIPCHK(hipGraphInstantiate(&g_solve, g, nullptr, nullptr, 0));
                (void)hipGraphDestroy(g);
            }
            hipLaunchKernelGGL(k_load_rhs, dim3(grid1d(S->n)), dim3(256), 0, stream, V, src);
            HIPCHK(hipGraphLaunch(g_solve, stream));
            hipLaunchKernelGGL(k_store_sol, dim3(grid1d(S->n)), dim3(256), 0, stream, V, col);
            return true;
        }
        return enqueue_solve(src, col);
    }
    bool solve_device(int nrhs, const double* dsrc, int lds_, double* drhs, int ld, bool timed) {
        DeviceGuard guard(dev);
        if (!ready) { if (err_.empty()) err_ = "solve: solver not set up"; return false; }
        if (multi) return solve_dist(nrhs, dsrc, lds_, drhs, ld);
        if (timed) HIPCHK(hipEventRecord(ev0, stream));
        int nctx = (nrhs > 1 && !prof_on && !V.strace && !knob_disabled("solve_ctx")) ? std::min(nrhs, SOLVE_CTX) : 1;
        if (nctx > 1 && sctx_verdict == 0 && !sctx_tuning) {
            // measured ONCE per structure, on scratch vectors: three right-hand sides one after the other against the same three through the contexts.  On a
            // tree whose chain sweeps fill the device (synth_1e6: 5 290 spinning workgroups) the level launches of a second context starve next to them and the
            // contexts LOSE (18.6 against 16.6 ms for eight right-hand sides); on a mid-size tree they win (grid_1e5: 2.2 against 3.5 ms).
            double* scratch = nullptr;
            HIPCHK(hipMalloc((void**)&scratch, 6 * (size_t)std::max(S->n, 1) * sizeof(double)));
            HIPCHK(hipMemsetAsync(scratch, 0, 6 * (size_t)std::max(S->n, 1) * sizeof(double), stream));
            sctx_tuning = true;
            float t_seq = 0, t_ctx = 0; bool ok = true;
            hipEvent_t e0 = nullptr, e1 = nullptr; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
            for (int mode = 0; mode < 2 && ok; ++mode) {
                sctx_verdict = mode == 0 ? 2 : 1;
                float best = 1e30f;
                for (int rep = 0; rep < 4 && ok; ++rep) {      // (the first round warms the contexts up: allocation, graph capture; then the best of three)
                    HIPCHK(hipEventRecord(e0, stream));
                    ok = solve_device(3, scratch, S->n, scratch + 3 * (size_t)S->n, S->n, false);
                    HIPCHK(hipEventRecord(e1, stream)); HIPCHK(hipStreamSynchronize(stream));
                    float ms = 0; if (ok) HIPCHK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep > 0) best = std::min(best, ms);
                }
                (mode == 0 ? t_seq : t_ctx) = best;
            }
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(scratch);
            sctx_tuning = false;
            if (!ok) { sctx_verdict = 0; return false; }
            sctx_verdict = t_ctx < 0.95f * t_seq ? 1 : 2;
            if (opt.verbose) fprintf(stderr, "[mi355x_kkt] solve contexts for nrhs > 1: three right-hand sides %.3f ms one after the other, %.3f ms through the contexts -> %s\n", t_seq, t_ctx, sctx_verdict == 1 ? "contexts" : "one after the other");
            if (timed) HIPCHK(hipEventRecord(ev0, stream));
        }
        if (sctx_verdict == 2 && !knob_int("solve_ctx_force", 0)) nctx = 1;
        if (nctx > 1) {
            if (!sctx_ensure(nctx)) return false;
            HIPCHK(hipEventRecord(sctx_fork, stream));
            for (int c = 1; c < nctx; ++c) HIPCHK(hipStreamWaitEvent(sctx[c].strm, sctx_fork, 0));
            gate_on = true; gate_valid = false;
            bool ok = true;
            for (int r = 0; r < nrhs && ok; ++r) {
                const int c = r % nctx;
                double* col = drhs + (size_t)r * ld;
                const double* src = dsrc + (size_t)r * lds_;
                gate_mine = gate_ev[c]; gate_entered = false;
                SctxSaved sv;
                if (c > 0) sctx_enter(c, sv);
                ok = enqueue_solve(src, col);          // eager: the gate of the chain sections is an event between the contexts' streams
                if (c > 0) sctx_leave(c, sv);
            }
            gate_on = false; gate_mine = nullptr;
            if (!ok) return false;
            for (int c = 1; c < nctx; ++c) {
                HIPCHK(hipEventRecord(sctx[c].done, sctx[c].strm)); HIPCHK(hipStreamWaitEvent(stream, sctx[c].done, 0));
                if (!chain_segs.empty()) hipLaunchKernelGGL(k_merge_err, dim3(1), dim3(64), 0, stream, V.sepoch + 1, (const int*)(sctx[c].sepoch + 1));
            }
        } else
            for (int r = 0; r < nrhs; ++r) if (!solve_one(dsrc + (size_t)r * lds_, drhs + (size_t)r * ld)) return false;
        if (timed) {
            HIPCHK(hipEventRecord(ev1, stream));
            if (!chain_segs.empty()) HIPCHK(hipMemcpyAsync(h_stats + 6, V.sepoch + 1, sizeof(int), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ev0, ev1)); solve_ms = ms;
            if (V.strace) {
                std::vector<unsigned long long> h(strace_n);
                HIPCHK(hipMemcpy(h.data(), V.strace, strace_n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                std::string trace_path; knob_trace("solve", &trace_path);
                if (FILE* f = fopen(trace_path.c_str(), "w")) {
                    for (size_t i = 0; i < strace_desc.size(); ++i)
                        fprintf(f, "%c %d %d %d %d %llu %llu %llu %llu\n", i < (size_t)V.strace_b ? 'F' : 'B', strace_desc[i].chain, strace_desc[i].w, strace_desc[i].nlinks, strace_desc[i].tail,
                                h[4 * i], h[4 * i + 1], h[4 * i + 2], h[4 * i + 3]);
                    fclose(f);
                }
            }
            if (!chain_segs.empty() && h_stats[6] != 0) { err_ = "solve: a chain sweep timed out waiting for its predecessor (workgroups not co-resident?)"; return false; }
        }
        return true;
    }


    // host copy between the caller's vectors and the pinned staging buffer: a single thread moves ~10 GB/s, a Solve at n = 10^6 moves
    // 2 x 32 MB -- the largest single item of the device route's PDSystemSolverTotal -- so large copies are split over a few threads
    static void par_memcpy(void* dst, const void* src, size_t bytes) {
        const size_t chunk = 4u << 20;
        const int nt = (int)std::min<size_t>(8, bytes / chunk);
        if (nt <= 1) { std::memcpy(dst, src, bytes); return; }
        std::vector<std::thread> th;
        const size_t per = ((bytes / nt) + 63) & ~(size_t)63;
        for (int t = 0; t < nt; ++t) {
            const size_t o = (size_t)t * per; if (o >= bytes) break;
            const size_t len = std::min(per, bytes - o);
            th.emplace_back([=]() { std::memcpy((char*)dst + o, (const char*)src + o, len); });
        }
        for (auto& x : th) x.join();
    }
    // ---------------- primal-dual (8-block) workspace: SURVEY 8(f)2 ----------------
    PdView pd_{}; bool pd_ready = false; long long pd_len8 = 0; int pd_dim4 = 0;
    static constexpr int PD_NVEC = 4;                      // RHS, RES, RESID, and one spare (the caller's copy)
    double* pd_vec[PD_NVEC] = {nullptr, nullptr, nullptr, nullptr};
    double* pd_aug = nullptr; double* pd_data = nullptr; int* pd_idx = nullptr; int* pd_rv = nullptr; double* pd_stage = nullptr;
    unsigned long long* pd_norms_d = nullptr; unsigned long long* pd_norms_h = nullptr;
    void pd_free() {
        for (int q = 0; q < PD_NVEC; ++q) if (pd_vec[q]) { (void)hipFree(pd_vec[q]); pd_vec[q] = nullptr; }
        if (pd_aug) { (void)hipFree(pd_aug); pd_aug = nullptr; }
        if (pd_data) { (void)hipFree(pd_data); pd_data = nullptr; }
        if (pd_idx) { (void)hipFree(pd_idx); pd_idx = nullptr; }
        if (pd_rv) { (void)hipFree(pd_rv); pd_rv = nullptr; }
        if (pd_stage) { (void)hipHostFree(pd_stage); pd_stage = nullptr; }
        if (pd_norms_d) { (void)hipFree(pd_norms_d); pd_norms_d = nullptr; }
        if (pd_norms_h) { (void)hipHostFree(pd_norms_h); pd_norms_h = nullptr; }
        pd_ready = false;
    }
    // dims = {nx, ns, nc, nd, nxl, nxu, nsl, nsu}; idx*: 0-based positions of the bounded entries; (irn, jcn): the 1-based triplets of
    // analyse(); segs: the assembly segments that hold W, J_c, J_d (everything else -- diagonals, -I -- is explicit in the kernels)
    bool pd_define(const int* dims, const int* ixl, const int* ixu, const int* isl, const int* isu, const int* irn, const int* jcn, const int* segs, int nsegs) {
        DeviceGuard guard(dev);
        if (!ready || asm_.nseg == 0) { err_ = "pd_define: analyse() and assembly_define() first"; return false; }
        pd_free();
        PdView& P = pd_;
        P.nx = dims[0]; P.ns = dims[1]; P.nc = dims[2]; P.nd = dims[3]; P.nxl = dims[4]; P.nxu = dims[5]; P.nsl = dims[6]; P.nsu = dims[7];
        pd_dim4 = P.nx + P.ns + P.nc + P.nd;
        if (pd_dim4 != S->n || P.ns != P.nd) { err_ = "pd_define: block dimensions do not match the analysed system"; return false; }
        pd_len8 = (long long)pd_dim4 + P.nxl + P.nxu + P.nsl + P.nsu;
        // row view of the selected segments, both triangles, entries of a row in triplet (slot) order
        std::vector<int> cnt(pd_dim4 + 1, 0);
        for (int q = 0; q < nsegs; ++q) {
            const int sg = segs[q];
            if (sg < 0 || sg >= asm_.nseg) { err_ = "pd_define: no such segment"; return false; }
            for (long long t = asm_.off[sg]; t < asm_.off[sg] + asm_.len[sg]; ++t) {
                const int r = irn[t] - 1, c = jcn[t] - 1;
                if (r < 0 || c < 0 || r >= pd_dim4 || c >= pd_dim4) { err_ = "pd_define: triplet index out of range"; return false; }
                cnt[r + 1]++; if (r != c) cnt[c + 1]++;
            }
        }
        for (int i = 0; i < pd_dim4; ++i) cnt[i + 1] += cnt[i];
        const int nent = cnt[pd_dim4];
        std::vector<int> rv((size_t)pd_dim4 + 1 + 2 * (size_t)nent), fill(cnt.begin(), cnt.end() - 1);
        std::copy(cnt.begin(), cnt.end(), rv.begin());
        int* rcol = rv.data() + pd_dim4 + 1; int* rslot = rcol + nent;
        std::vector<int> order(segs, segs + nsegs);
        std::sort(order.begin(), order.end(), [&](int a, int b) { return asm_.off[a] < asm_.off[b]; });
        for (int sg : order)
            for (long long t = asm_.off[sg]; t < asm_.off[sg] + asm_.len[sg]; ++t) {
                const int r = irn[t] - 1, c = jcn[t] - 1;
                rcol[fill[r]] = c; rslot[fill[r]++] = (int)t;
                if (r != c) { rcol[fill[c]] = r; rslot[fill[c]++] = (int)t; }
            }
        HIPCHK(hipMalloc((void**)&pd_rv, rv.size() * sizeof(int)));
        HIPCHK(hipMemcpy(pd_rv, rv.data(), rv.size() * sizeof(int), hipMemcpyHostToDevice));
        P.rptr = pd_rv; P.rcol = pd_rv + pd_dim4 + 1; P.rslot = P.rcol + nent;
        const long long nb = (long long)P.nxl + P.nxu + P.nsl + P.nsu;
        HIPCHK(hipMalloc((void**)&pd_idx, std::max<long long>(nb, 1) * sizeof(int)));
        HIPCHK(hipMalloc((void**)&pd_data, std::max<long long>(2 * nb, 1) * sizeof(double)));
        {
            int* d = pd_idx;
            P.ixl = d; if (P.nxl) HIPCHK(hipMemcpy(d, ixl, P.nxl * sizeof(int), hipMemcpyHostToDevice)); d += P.nxl;
            P.ixu = d; if (P.nxu) HIPCHK(hipMemcpy(d, ixu, P.nxu * sizeof(int), hipMemcpyHostToDevice)); d += P.nxu;
            P.isl = d; if (P.nsl) HIPCHK(hipMemcpy(d, isl, P.nsl * sizeof(int), hipMemcpyHostToDevice)); d += P.nsl;
            P.isu = d; if (P.nsu) HIPCHK(hipMemcpy(d, isu, P.nsu * sizeof(int), hipMemcpyHostToDevice));
            double* f = pd_data;
            P.zl = f; f += P.nxl; P.zu = f; f += P.nxu; P.vl = f; f += P.nsl; P.vu = f; f += P.nsu;
            P.sxl = f; f += P.nxl; P.sxu = f; f += P.nxu; P.ssl = f; f += P.nsl; P.ssu = f;
        }
        for (int q = 0; q < PD_NVEC; ++q) { HIPCHK(hipMalloc((void**)&pd_vec[q], std::max<long long>(pd_len8, 1) * sizeof(double))); HIPCHK(hipMemset(pd_vec[q], 0, std::max<long long>(pd_len8, 1) * sizeof(double))); }
        HIPCHK(hipMalloc((void**)&pd_aug, std::max(pd_dim4, 1) * sizeof(double)));
        HIPCHK(hipHostMalloc((void**)&pd_stage, std::max<long long>(std::max<long long>(pd_len8, 2 * nb), 1) * sizeof(double), hipHostMallocDefault));
        HIPCHK(hipMalloc((void**)&pd_norms_d, 4 * sizeof(unsigned long long)));
        HIPCHK(hipHostMalloc((void**)&pd_norms_h, 4 * sizeof(unsigned long long), hipHostMallocDefault));
        P.norms = pd_norms_d; P.tvals = V.tvals;
        HIPCHK(hipDeviceSynchronize());          // (the zero fills above ran on the default stream)
        pd_ready = true;
        return true;
    }
    // the iterate's bound multipliers and slacks: arr = {z_L, z_U, v_L, v_U, slack_x_L, slack_x_U, slack_s_L, slack_s_U}
    bool pd_put_data(const double* const* arr) {
        DeviceGuard guard(dev);
        if (!pd_ready) { err_ = "pd_put_data: pd_define first"; return false; }
        const int len[8] = {pd_.nxl, pd_.nxu, pd_.nsl, pd_.nsu, pd_.nxl, pd_.nxu, pd_.nsl, pd_.nsu};
        HIPCHK(hipStreamSynchronize(stream));                  // the staging buffer may still feed an earlier copy
        long long o = 0;
        for (int q = 0; q < 8; ++q) { if (len[q]) par_memcpy(pd_stage + o, arr[q], (size_t)len[q] * sizeof(double)); o += len[q]; }
        if (o) HIPCHK(hipMemcpyAsync(pd_data, pd_stage, (size_t)o * sizeof(double), hipMemcpyHostToDevice, stream));
        return true;
    }
    int pd_blocklen(int b) const { const int len[8] = {pd_.nx, pd_.ns, pd_.nc, pd_.nd, pd_.nxl, pd_.nxu, pd_.nsl, pd_.nsu}; return len[b]; }
    bool pd_put(int vec, const double* const* blocks) {
        DeviceGuard guard(dev);
        if (!pd_ready || vec < 0 || vec >= PD_NVEC) { err_ = "pd_put: pd_define first / no such vector"; return false; }
        HIPCHK(hipStreamSynchronize(stream));
        long long o = 0;
        for (int b = 0; b < 8; ++b) { const int l = pd_blocklen(b); if (l) par_memcpy(pd_stage + o, blocks[b], (size_t)l * sizeof(double)); o += l; }
        if (o) HIPCHK(hipMemcpyAsync(pd_vec[vec], pd_stage, (size_t)o * sizeof(double), hipMemcpyHostToDevice, stream));
        return true;
    }
    bool pd_get(int vec, double* const* blocks) {
        DeviceGuard guard(dev);
        if (!pd_ready || vec < 0 || vec >= PD_NVEC) { err_ = "pd_get: pd_define first / no such vector"; return false; }
        if (pd_len8) HIPCHK(hipMemcpyAsync(pd_stage, pd_vec[vec], (size_t)pd_len8 * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        long long o = 0;
        for (int b = 0; b < 8; ++b) { const int l = pd_blocklen(b); if (l) par_memcpy(blocks[b], pd_stage + o, (size_t)l * sizeof(double)); o += l; }
        return true;
    }
    // res <- alpha sol + beta res,  sol = the solution of the 8-block system with right-hand side `rhs` through the CURRENT factorisation
    // of the augmented system (reduce, 4-block solve, expand: SolveOnce without its inertia-correction loop, which stays with the caller)
    bool pd_solve_once(int rhs, int res, double alpha, double beta) {
        DeviceGuard guard(dev);
        if (!pd_ready || rhs < 0 || rhs >= PD_NVEC || res < 0 || res >= PD_NVEC) { err_ = "pd_solve_once: pd_define first / no such vector"; return false; }
        const int g4 = grid1d(pd_dim4), gb = grid1d(std::max(std::max(pd_.nxl, pd_.nxu), std::max(pd_.nsl, pd_.nsu)));
        hipLaunchKernelGGL(k_pd_reduce, dim3(g4), dim3(256), 0, stream, pd_, (const double*)pd_vec[rhs], pd_aug, 0);
        hipLaunchKernelGGL(k_pd_reduce, dim3(gb), dim3(256), 0, stream, pd_, (const double*)pd_vec[rhs], pd_aug, 1);
        hipLaunchKernelGGL(k_pd_reduce, dim3(gb), dim3(256), 0, stream, pd_, (const double*)pd_vec[rhs], pd_aug, 2);
        if (!solve_device(1, pd_aug, pd_dim4, pd_aug, pd_dim4, false)) return false;
        hipLaunchKernelGGL(k_pd_expand, dim3(g4), dim3(256), 0, stream, pd_, (const double*)pd_vec[rhs], (const double*)pd_aug, pd_vec[res], alpha, beta);
        HIPCHK(hipGetLastError());
        return true;
    }
    // resid <- residual of the unreduced system at `res`; norms = {|rhs|_inf, |res|_inf, |resid|_inf}
    bool pd_residual(int rhs, int res, int resid, const double* deltas, double* norms) {
        DeviceGuard guard(dev);
        if (!pd_ready || rhs < 0 || rhs >= PD_NVEC || res < 0 || res >= PD_NVEC || resid < 0 || resid >= PD_NVEC) { err_ = "pd_residual: pd_define first / no such vector"; return false; }
        const int g4 = grid1d(pd_dim4), gb = grid1d(std::max(std::max(pd_.nxl, pd_.nxu), std::max(pd_.nsl, pd_.nsu)));
        hipLaunchKernelGGL(k_zero_u64, dim3(1), dim3(64), 0, stream, pd_norms_d, 4);
        hipLaunchKernelGGL(k_pd_resid_rows, dim3(g4), dim3(256), 0, stream, pd_, (const double*)pd_vec[rhs], (const double*)pd_vec[res], pd_vec[resid], deltas[0], deltas[1], deltas[2], deltas[3]);
        hipLaunchKernelGGL(k_pd_resid_bounds, dim3(gb), dim3(256), 0, stream, pd_, (const double*)pd_vec[rhs], (const double*)pd_vec[res], pd_vec[resid], 1);
        hipLaunchKernelGGL(k_pd_resid_bounds, dim3(gb), dim3(256), 0, stream, pd_, (const double*)pd_vec[rhs], (const double*)pd_vec[res], pd_vec[resid], 2);
        hipLaunchKernelGGL(k_pd_norms, dim3(grid1d(pd_len8)), dim3(256), 0, stream, pd_, (const double*)pd_vec[rhs], (const double*)pd_vec[res], (const double*)pd_vec[resid], pd_len8);
        HIPCHK(hipMemcpyAsync(pd_norms_h, pd_norms_d, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        for (int q = 0; q < 3; ++q) { double v; std::memcpy(&v, &pd_norms_h[q], sizeof v); norms[q] = v; }
        return true;
    }

    // ---------------- multi-GPU orchestration (eager launches) ----------------
    bool launch_fronts(const Sched& sc, int top_mode) {
        const Symbolic& Sy = *S;
        for (int lv = 0; lv < Sy.num_levels; ++lv)
            for (int fc = 0; fc < FC_COUNT; ++fc) {
                const int b0 = sc.base + sc.ptr[(size_t)lv * FC_COUNT + fc], b1 = sc.base + sc.ptr[(size_t)lv * FC_COUNT + fc + 1];
                if (b1 == b0) continue;
                launch_bucket(lv, fc, b0, b1, top_mode, sc.maxm[lv], sc.maxk[lv], sc.tiles[lv], sc.tiles64[lv]);
            }
        HIPCHK(hipGetLastError());
        return true;
    }
    bool launch_solve_sweep(const Sched& sc, bool forward, int top_mode, bool use_segs = true) {
        const Symbolic& Sy = *S;
        auto lds_solve = [](int mmax, int kmax) { return (size_t)(mmax + 3 * kmax) * sizeof(double) + 16; };
        for (int q = 0; q < Sy.num_levels; ++q) {
            const int lv = forward ? q : Sy.num_levels - 1 - q;
            if (top_mode && use_segs && !chain_segs.empty()) {      // runs of pure chain levels of the fronts every rank holds: the sync-free sweeps (pure links have no child from outside)
                const int sgi = forward ? seg_at_lv0[lv] : seg_at_lv1[lv];
                if (sgi >= 0) {
                    const ChainSeg& sg = chain_segs[sgi];
                    if (forward) hipLaunchKernelGGL(k_fwd_chain, dim3(sg.nwg_f), dim3(320), 0, stream, V, sg.wgf0);
                    else hipLaunchKernelGGL(k_bwd_chain, dim3(sg.nwg_b), dim3(320), 0, stream, V, sg.wgb0);
                    q += sg.lv1 - sg.lv0; continue;
                }
            }
            for (int fc = 0; fc < FC_COUNT; ++fc) {
                const int b0 = sc.base + sc.ptr[(size_t)lv * FC_COUNT + fc], b1 = sc.base + sc.ptr[(size_t)lv * FC_COUNT + fc + 1];
                if (b1 == b0) continue;
                if (forward) {
                    if (fc == FC_WAVE)        hipLaunchKernelGGL((k_fwd<64>),  dim3(b1 - b0), dim3(64),  lds_solve(32, 32),   stream, V, b0, top_mode);
                    else if (fc == FC_LDS64)  hipLaunchKernelGGL((k_fwd<64>),  dim3(b1 - b0), dim3(64),  lds_solve(64, 64),   stream, V, b0, top_mode);
                    else if (fc == FC_LDS128) hipLaunchKernelGGL((k_fwd<256>), dim3(b1 - b0), dim3(256), lds_solve(128, 128), stream, V, b0, top_mode);
                    else if (sc.last1[lv] > sc.last0[lv]) {
                           const int g0 = sc.last0[lv], ng = sc.last1[lv] - g0;
                           if (!top_mode && sc.allsolo[lv]) hipLaunchKernelGGL(k_fwd_solo, dim3((sc.maxm[lv] + 63) / 64 + 1, ng), dim3(256), 0, stream, V, g0);
                           else { hipLaunchKernelGGL(k_fwd_grp, dim3(ng), dim3(256), 0, stream, V, g0, top_mode);
                                  hipLaunchKernelGGL(k_fwd_grp_upd, dim3((sc.maxm[lv] + 63) / 64, ng), dim3(256), 0, stream, V, g0); } }
                } else {
                    if (fc == FC_WAVE)        hipLaunchKernelGGL((k_bwd<64>),  dim3(b1 - b0), dim3(64),  lds_solve(32, 32),   stream, V, b0);
                    else if (fc == FC_LDS64)  hipLaunchKernelGGL((k_bwd<64>),  dim3(b1 - b0), dim3(64),  lds_solve(64, 64),   stream, V, b0);
                    else if (fc == FC_LDS128) hipLaunchKernelGGL((k_bwd<256>), dim3(b1 - b0), dim3(256), lds_solve(128, 128), stream, V, b0);
                    else if (sc.last1[lv] > sc.last0[lv]) {
                           const int g0 = sc.last0[lv], ng = sc.last1[lv] - g0;
                           hipLaunchKernelGGL(k_bwd_grp_dot, dim3((sc.maxm[lv] + 255) / 256, ng), dim3(256), 0, stream, V, g0);
                           hipLaunchKernelGGL(k_bwd_grp, dim3(ng), dim3(256), 0, stream, V, g0); }
                }
            }
        }
        HIPCHK(hipGetLastError());
        return true;
    }
    // the phase entry points (the caller runs the collectives between them: ipopt_amd.multigpu.DistributedKKT) exist for the classic mapping,
    // ONE exchange step; with the subtree-to-subcube mapping the sequence runs behind the ordinary entry points (factor_dist / solve_dist)
    bool one_step(const char* who) { if (ndepth != 1) { err_ = std::string(who) + ": the phase entry points serve one exchange step (subcube = 0); set a communicator and use factor / solve"; return false; } return true; }
    bool factor_local(const double* dvals) {
        DeviceGuard guard(dev);
        if (!one_step("factor_local")) return false;
        if (!enqueue_factor_local(dvals, false)) return false;
        HIPCHK(hipEventRecord(ev1, stream));
        HIPCHK(hipStreamSynchronize(stream));       // the caller's collective runs on another stream
        float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ev0, ev1)); factor_ms = ms;
        return true;
    }
    bool report_to_arena(int kind) {
        const JoinList& J = join[kind];
        if (J.count > 0) hipLaunchKernelGGL(k_arena_assemble, dim3((J.maxm + 3) / 4, J.count), dim3(256), 0, stream, V, J.base, J.who);
        HIPCHK(hipGetLastError());
        return true;
    }
    bool report_to_top_rhs(int kind) {
        const JoinList& J = join[kind];
        if (J.count > 0) hipLaunchKernelGGL(k_top_rhs_assemble, dim3(J.count), dim3(256), 0, stream, V, J.base, J.who);
        HIPCHK(hipGetLastError());
        return true;
    }
    // own subtrees + own contributions to the arena squares above them, enqueued on the solver's stream (no host synchronisation)
    bool enqueue_factor_local(const double* dvals, bool reuse) {
        if (!ready || !multi) { err_ = "factor_local: not a multi-GPU handle (nranks must be > 1 at create)"; return false; }
        const Symbolic& Sy = *S; const int n = Sy.n;
        V.pivtol = opt.pivtol; V.pivtol2 = std::max(opt.pivtol, opt.pivtolmax); V.small = opt.small;
        if (!reuse) {
            if (dvals) HIPCHK(hipMemcpyAsync((void*)V.tvals, dvals, Sy.nnz_in * sizeof(double), hipMemcpyDeviceToDevice, stream));
            else       HIPCHK(hipMemcpyAsync((void*)V.tvals, h_vals, Sy.nnz_in * sizeof(double), hipMemcpyHostToDevice, stream));
            have_values = true;
        } else if (!have_values) { err_ = "refactor: no values on the device yet"; return false; }
        if (want_matching() && n > 0) { if (!run_matching()) return false; match_valid = true; }
        HIPCHK(hipEventRecord(ev0, stream));
        hipLaunchKernelGGL(k_factor_prologue, dim3(grid1d(n)), dim3(256), 0, stream, V);
        hipLaunchKernelGGL(k_gather_values, dim3(grid1d(Sy.nnz_a)), dim3(256), 0, stream, V);
        enqueue_scaling();
        if (!launch_fronts(sch_local, 0)) return false;
        HIPCHK(hipMemsetAsync(V.arena, 0, (size_t)arena_doubles * sizeof(double), stream));
        return report_to_arena(0);
    }
    // replicated fronts of exchange step d held by this rank (their arena squares have been summed); afterwards the first rank of the range
    // reports what they contribute to the fronts of the wider ranges above
    bool enqueue_factor_step(int d) {
        gs_cur = &gs_stage[d];
        if (!launch_fronts(sch_stage[d], 1)) return false;
        return d > 0 ? report_to_arena(1 + d) : true;
    }
    bool enqueue_stats() {
        // this rank counts its own subtrees and the replicated fronts whose range it is the first rank of => the sum over ranks is the inertia
        hipLaunchKernelGGL(k_zero_i32, dim3(1), dim3(64), 0, stream, d_stats, 4);
        hipLaunchKernelGGL(k_reduce_stats, dim3(1), dim3(256), 0, stream, V.fstat, V.apfail, V.sn_owner, S->num_sn, opt.rank, d_stats);
        hipLaunchKernelGGL(k_reduce_stats, dim3(1), dim3(256), 0, stream, V.fstat, V.apfail, V.sn_owner, S->num_sn, -1, d_stats);
        HIPCHK(hipGetLastError());
        return true;
    }
    // The whole distributed factorisation behind the ordinary factor() entry point (a communicator has been set):
    //   own subtrees -> per exchange step, deepest ranges of ranks first: all-reduce(that step's arena squares) -> the step's replicated
    //   fronts -> ... -> all-reduce(inertia / pivot statistics),
    // everything stream-ordered on the solver's stream, ONE host synchronisation at the end.  Every rank takes part in every collective
    // (with zeros for the squares of ranges it is not in), in the same order: no sub-communicators, nothing to deadlock.
    bool factor_dist(const double* dvals, bool reuse, FactorStats& st) {
        if (!enqueue_factor_local(dvals, reuse)) return false;
        for (int d = ndepth - 1; d >= 0; --d) {
            if (!exchange_step(d, 0)) return false;
            if (!enqueue_factor_step(d)) return false;
        }
        if (!enqueue_stats()) return false;
        if (!allreduce(d_stats, 8, 1)) return false;
        HIPCHK(hipEventRecord(ev1, stream));
        HIPCHK(hipMemcpyAsync(h_stats, d_stats, 8 * sizeof(int), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ev0, ev1)); factor_ms = ms;
        st.num_neg = h_stats[0]; st.num_zero = h_stats[1]; st.num_two = h_stats[2]; st.num_small = h_stats[3]; st.u_sensitive = h_stats[4] != 0; st.num_fast = h_stats[7];
        if (h_stats[5] != 0) { err_ = "factor: a panel workgroup timed out waiting for its pivot block on some rank (workgroups not co-resident?)"; return false; }      // (a sum over the ranks)
        return true;
    }
    bool enqueue_fwd_local(const double* src) {
        hipLaunchKernelGGL(k_load_rhs, dim3(grid1d(S->n)), dim3(256), 0, stream, V, src);
        if (!chain_segs.empty()) hipLaunchKernelGGL(k_bump_epoch, dim3(1), dim3(64), 0, stream, V.sepoch);
        if (toprhs_doubles > 0) HIPCHK(hipMemsetAsync(V.top_rhs, 0, (size_t)toprhs_doubles * sizeof(double), stream));
        if (!launch_solve_sweep(sch_local, true, 0)) return false;
        return report_to_top_rhs(0);
    }
    // distributed solve of one right-hand side (identical on every rank), solution on every rank:
    //   local forward -> per exchange step, deepest first: all-reduce(that step's top right-hand sides) -> forward on the step's fronts ->
    //   backward on the replicated fronts, widest range first (a front's ancestors are all held by its ranks: nothing to exchange) ->
    //   local backward -> all-reduce of the solution pieces
    bool solve_dist(int nrhs, const double* dsrc, int lds_, double* drhs, int ld) {
        HIPCHK(hipEventRecord(ev0, stream));
        for (int r = 0; r < nrhs; ++r) {
            const double* src = dsrc + (size_t)r * lds_; double* col = drhs + (size_t)r * ld;
            if (!enqueue_fwd_local(src)) return false;
            for (int d = ndepth - 1; d >= 0; --d) {
                if (!exchange_step(d, 1)) return false;
                if (!launch_solve_sweep(sch_stage[d], true, 1, d == 0)) return false;
                if (d > 0 && !report_to_top_rhs(1 + d)) return false;
            }
            for (int d = 0; d < ndepth; ++d) if (!launch_solve_sweep(sch_stage[d], false, 1, d == 0)) return false;
            if (!launch_solve_sweep(sch_local, false, 0)) return false;
            hipLaunchKernelGGL(k_store_sol_mg, dim3(grid1d(S->n)), dim3(256), 0, stream, V, col);
            HIPCHK(hipGetLastError());
            if (!allreduce(col, S->n, 0)) return false;
        }
        HIPCHK(hipEventRecord(ev1, stream));
        if (!chain_segs.empty()) HIPCHK(hipMemcpyAsync(h_stats + 6, V.sepoch + 1, sizeof(int), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ev0, ev1)); solve_ms = ms;
        if (!chain_segs.empty() && h_stats[6] != 0) { err_ = "solve: a chain sweep timed out waiting for its predecessor on this rank (workgroups not co-resident: several ranks on one GPU?)"; return false; }
        return true;
    }
    bool factor_top(FactorStats& st) {
        DeviceGuard guard(dev);
        if (!ready || !multi) { err_ = "factor_top: not a multi-GPU handle"; return false; }
        if (!one_step("factor_top")) return false;
        HIPCHK(hipEventRecord(ev0, stream));
        if (!enqueue_factor_step(0) || !enqueue_stats()) return false;
        HIPCHK(hipMemcpyAsync(h_stats, d_stats, 8 * sizeof(int), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipEventRecord(ev1, stream)); HIPCHK(hipStreamSynchronize(stream));
        st.num_neg = h_stats[0]; st.num_zero = h_stats[1]; st.num_two = h_stats[2]; st.num_small = h_stats[3]; st.u_sensitive = h_stats[4]; st.num_fast = h_stats[7];
        float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ev0, ev1)); factor_ms += ms;
        return true;
    }
    bool solve_fwd_local(double* drhs) {
        DeviceGuard guard(dev);
        if (!ready || !multi) { err_ = "solve_fwd_local: not a multi-GPU handle"; return false; }
        if (!one_step("solve_fwd_local")) return false;
        HIPCHK(hipEventRecord(ev0, stream));
        if (!enqueue_fwd_local((const double*)drhs)) return false;
        HIPCHK(hipEventRecord(ev1, stream)); HIPCHK(hipStreamSynchronize(stream));
        float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ev0, ev1)); solve_ms = ms;
        return true;
    }
    bool solve_top_and_bwd(double* drhs) {
        DeviceGuard guard(dev);
        if (!ready || !multi) { err_ = "solve_top_and_bwd: not a multi-GPU handle"; return false; }
        if (!one_step("solve_top_and_bwd")) return false;
        HIPCHK(hipEventRecord(ev0, stream));
        if (!launch_solve_sweep(sch_stage[0], true, 1)) return false;
        if (!launch_solve_sweep(sch_stage[0], false, 1)) return false;
        if (!launch_solve_sweep(sch_local, false, 0)) return false;
        hipLaunchKernelGGL(k_store_sol_mg, dim3(grid1d(S->n)), dim3(256), 0, stream, V, drhs);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(ev1, stream)); HIPCHK(hipStreamSynchronize(stream));
        float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ev0, ev1)); solve_ms += ms;
        return true;
    }

    // indices (original numbering, 0-based) of the columns whose pivot was (numerically) zero in the last factorisation
    bool zero_pivots(std::vector<int>& out) {
        DeviceGuard guard(dev);
        out.clear();
        if (!ready) { err_ = "zero_pivots: solver not set up"; return false; }
        std::vector<int> z(S->n);
        if (S->n > 0) HIPCHK(hipMemcpy(z.data(), V.zpiv, (size_t)S->n * sizeof(int), hipMemcpyDeviceToHost));
        for (int i = 0; i < S->n; ++i) if (z[i] & 1) out.push_back(S->perm[i]);      // (bit 1: delayed-pivot mark, failed_pivots)
        std::sort(out.begin(), out.end());
        return true;
    }
    // the columns (CURRENT permuted numbering) the last factorisation could not pivot and eliminated by static pivoting: the
    // delayed pivots of a solver with dynamic fronts (bit 1 of zpiv: ldlt_reg when every candidate of a front had failed, k_big_trsm a posteriori).
    // Multi-GPU: every rank sees the marks of the fronts it factored; the union is formed through the communicator (same list on every rank).
    bool failed_pivots(std::vector<int>& out) {
        DeviceGuard guard(dev);
        out.clear();
        if (!ready) { err_ = "failed_pivots: solver not set up"; return false; }
        const int n = S->n;
        std::vector<int> z(std::max(n, 1));
        if (n > 0) HIPCHK(hipMemcpy(z.data(), V.zpiv, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
        if (multi && comm_kind != 0 && n > 0) {
            for (int i = 0; i < n; ++i) z[i] = (z[i] >> 1) & 1;
            HIPCHK(hipMemcpyAsync(V.colfail, z.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, stream));
            if (!allreduce(V.colfail, n, 1)) return false;
            HIPCHK(hipMemcpyAsync(z.data(), V.colfail, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            for (int i = 0; i < n; ++i) if (z[i]) out.push_back(i);
        } else for (int i = 0; i < n; ++i) if (z[i] & 2) out.push_back(i);
        return true;
    }
    // development aid: D^{-1}, the 2x2 off-diagonals, pivot types and the within-front pivot order of the last factorisation (permuted numbering)
    bool debug_pivots(double* dinv, double* doff, int* ptype, int* lperm) {
        DeviceGuard guard(dev);
        if (!ready) return false;
        const size_t n = S->n;
        HIPCHK(hipMemcpy(dinv, V.dinv, n * sizeof(double), hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(doff, V.doff, n * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(ptype, V.ptype, n * sizeof(int), hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(lperm, V.lperm, n * sizeof(int), hipMemcpyDeviceToHost));
        return true;
    }
    bool debug_clocks(unsigned long long* out) {
        DeviceGuard guard(dev);
        if (!V.dbg) { err_ = "debug clocks not enabled (MI355X_KKT_TRACE=clocks)"; return false; }
        HIPCHK(hipMemcpy(out, V.dbg, 128 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
#ifdef MI355X_PIVSTAT
        unsigned long long ps[16]; HIPCHK(hipMemcpyFromSymbol(ps, HIP_SYMBOL(g_pivstat), sizeof ps));
        unsigned long long fs[32]; HIPCHK(hipMemcpyFromSymbol(fs, HIP_SYMBOL(g_fstat), sizeof fs));
        for (int o = 16; o <= 24; o += 8) { const double nf = (double)std::max(1ull, fs[o + 4]);
            fprintf(stderr, "PIVSTAT k_front_reg<256,%d>: %llu fronts, mean k %.1f, children %.1f; cycles per front: assembly %.0f, to registers %.0f, colmax+LDL^T %.0f, write-back+inverse %.0f\n", o == 16 ? 6 : 8, fs[o + 4], fs[o + 5] / nf, fs[o + 6] / nf, fs[o] / nf, fs[o + 1] / nf, fs[o + 2] / nf, fs[o + 3] / nf); }
        unsigned long long da[16]; HIPCHK(hipMemcpyFromSymbol(da, HIP_SYMBOL(g_dtacc), sizeof da));
        { const double nl = (double)std::max(1ull, da[0]) * 100.0;      // 100 MHz ticks -> us
          fprintf(stderr, "PIVSTAT fused pivot block + panel solve, one front per launch (%llu launches), us after the pivot workgroup started: tiles loaded %.1f, LDL^T done %.1f, permuted + written %.1f, inverse done %.1f (previous launch), flag about to be raised %.1f | panel workgroup 1: started %.1f, rows staged %.1f, flag seen %.1f, L11 + D loaded %.1f, permuted + diagonal blocks inverted %.1f, solved %.1f, done %.1f\n",
                  da[0], da[1] / nl, da[2] / nl, da[3] / nl, da[4] / nl, da[5] / nl, da[6] / nl, da[7] / nl, da[8] / nl, da[10] / nl, da[11] / nl, da[12] / nl, da[9] / nl); }
        fprintf(stderr, "PIVSTAT big pivot blocks: fast steps %llu mean %.0f cycles, slow steps %llu mean %.0f cycles\n", ps[10], (double)ps[8] / (double)std::max(1ull, ps[10]), ps[11], (double)ps[9] / (double)std::max(1ull, ps[11]));
        fprintf(stderr, "PIVSTAT slow-path |a_jj|/lambda: >=0.5 %llu  [0.25,0.5) %llu  [0.1,0.25) %llu  <0.1 %llu (of which <0.01 %llu)\n", ps[12], ps[13], ps[14], ps[15], ps[7]);
        fprintf(stderr, "PIVSTAT steps %llu slow %llu quick %llu exact %llu passover %llu twobytwo %llu nopartner %llu\n", ps[0], ps[1], ps[2], ps[3], ps[4], ps[5], ps[6]);
#endif
        return true;
    }
    // eager (graph-less) factor + one solve with hip events around every launch; accumulates over `reps`
    bool profile(int reps, double* ms, int* launches) {
        DeviceGuard guard(dev);
        if (!ready || !have_values) { err_ = "profile: factor() must have been called once"; return false; }
        for (int q = 0; q < KK_COUNT; ++q) { prof_ms[q] = 0; prof_launches[q] = 0; }
        if (d_rhs_cap < (size_t)S->n) { if (d_rhs) (void)hipFree(d_rhs); d_rhs = nullptr; HIPCHK(hipMalloc((void**)&d_rhs, std::max<size_t>(S->n, 1) * sizeof(double))); d_rhs_cap = S->n; HIPCHK(hipMemset(d_rhs, 0, S->n * sizeof(double))); HIPCHK(hipDeviceSynchronize());
                                        if (g_solve) { (void)hipGraphExecDestroy(g_solve); g_solve = nullptr; } }
        prof_on = true;
        for (int r = 0; r < reps; ++r) {
            if (!enqueue_factor() || !enqueue_solve(d_rhs, d_rhs)) { prof_on = false; return false; }
            prof_collect();
        }
        prof_on = false;
        for (int q = 0; q < KK_COUNT; ++q) { ms[q] = prof_ms[q]; launches[q] = prof_launches[q]; }
        return true;
    }
    bool solve_host(int nrhs, double* rhs, int ld) {
        DeviceGuard guard(dev);
        if (!ready) { if (err_.empty()) err_ = "solve: solver not set up"; return false; }
        const size_t n = S->n;
        // nrhs > 1 (IpLowRankAugSystemSolver.cpp:435-487, sIPOPT): ALL columns go up in one stream-ordered batch, the sweeps of the columns run back to back
        // on the device with no host synchronisation between them, all solutions come down behind the last one -- one PCIe round trip and one wait per
        // call instead of one per column (round 4: upload, solve, download, synchronise per column).  L is still streamed once per column: the sweeps are
        // written for one vector (panel rows in registers, 16-lane DPP rows); a blocked multi-vector version of them is not built.
        const size_t need = n * (size_t)std::max(nrhs, 1);
        if (d_rhs_cap < need) { if (d_rhs) (void)hipFree(d_rhs); d_rhs = nullptr; HIPCHK(hipMalloc((void**)&d_rhs, std::max<size_t>(need, 1) * sizeof(double))); d_rhs_cap = need;
                                if (g_solve) { (void)hipGraphExecDestroy(g_solve); g_solve = nullptr; } }
        if ((size_t)ld == n || nrhs == 1) HIPCHK(hipMemcpyAsync(d_rhs, rhs, n * (size_t)nrhs * sizeof(double), hipMemcpyHostToDevice, stream));
        else for (int r = 0; r < nrhs; ++r) HIPCHK(hipMemcpyAsync(d_rhs + (size_t)r * n, rhs + (size_t)r * ld, n * sizeof(double), hipMemcpyHostToDevice, stream));
        if (!solve_device(nrhs, d_rhs, (int)n, d_rhs, (int)n, true)) return false;          // (synchronises once, behind the last column; solve_ms = all columns)
        if ((size_t)ld == n || nrhs == 1) HIPCHK(hipMemcpyAsync(rhs, d_rhs, n * (size_t)nrhs * sizeof(double), hipMemcpyDeviceToHost, stream));
        else for (int r = 0; r < nrhs; ++r) HIPCHK(hipMemcpyAsync(rhs + (size_t)r * ld, d_rhs + (size_t)r * n, n * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return true;
    }
};

Numeric::Numeric() : p_(new NumericImpl) {}
Numeric::~Numeric() { delete p_; }
bool Numeric::setup(const Symbolic& S, const NumericOptions& opt) { return p_->setup(S, opt); }
double* Numeric::values_buffer() { return p_->h_vals; }
// the ordinal a handle created with `device` will use, resolved on the CALLER's thread (HIP's current device is per thread: a helper thread
// asking for "the current device" gets device 0 whatever the caller selected); -1: no usable device
int Numeric::resolve_device(int device)
{
    if (device >= 0) return device;
    int ndev = 0, dev = -1;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return -1;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    return dev;
}
void* Numeric::prewarm(int device, size_t count)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return nullptr;
    int dev = device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return nullptr;
    int prev = 0; (void)hipGetDevice(&prev);
    if (hipSetDevice(dev) != hipSuccess) return nullptr;
    (void)hipFree(nullptr);                                   // context
    hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, (const void*)k_bump_epoch);      // code object of this library
    void* p = nullptr;
    if (hipHostMalloc(&p, std::max<size_t>(count, 1) * sizeof(double), hipHostMallocDefault) != hipSuccess) p = nullptr;
    (void)hipSetDevice(prev);
    return p;
}
void Numeric::prewarm_discard(void* p) { if (p) (void)hipHostFree(p); }
bool Numeric::factor(const double* dvals, bool reuse, FactorStats& st) { return p_->factor(dvals, reuse, st); }
bool Numeric::solve_host(int nrhs, double* rhs, int ld) { return p_->solve_host(nrhs, rhs, ld); }
bool Numeric::solve_device(int nrhs, double* drhs, int ld) { return p_->solve_device(nrhs, drhs, ld, drhs, ld, true); }
bool Numeric::solve_device2(int nrhs, const double* db, int ldb, double* dx, int ldx) { return p_->solve_device(nrhs, db, ldb, dx, ldx, true); }
void Numeric::set_pivtol(double u) { if (u != p_->opt.pivtol) p_->scale_valid = false; p_->opt.pivtol = u; }      // (IncreaseQuality: the scaling is computed afresh, as MA97's rescale)
void Numeric::set_pivtolmax(double u) { p_->opt.pivtolmax = u; }
double Numeric::last_factor_ms() const { return p_->factor_ms; }
void Numeric::matching_stats(double* ms, int* rounds, int* unmatched) const { *ms = p_->match_ms; *rounds = p_->match_rounds; *unmatched = p_->match_unmatched; }
double Numeric::last_solve_ms() const { return p_->solve_ms; }
const std::string& Numeric::error() const { return p_->err_; }
bool Numeric::profile(int reps, double* ms, int* launches) { return p_->profile(reps, ms, launches); }
bool Numeric::debug_clocks(unsigned long long* out) { return p_->debug_clocks(out); }
bool Numeric::debug_pivots(double* a, double* b, int* c, int* d) { return p_->debug_pivots(a, b, c, d); }
bool Numeric::factor_local(const double* dvals) { return p_->factor_local(dvals); }
bool Numeric::top_arena(double** d, int64_t* nd) { if (!p_->multi) { p_->err_ = "top_arena: not a multi-GPU handle"; return false; } *d = p_->V.arena; *nd = p_->arena_doubles; return true; }
bool Numeric::factor_top(FactorStats& st) { return p_->factor_top(st); }
bool Numeric::solve_fwd_local(double* drhs) { return p_->solve_fwd_local(drhs); }
bool Numeric::top_rhs(double** d, int64_t* nd) { if (!p_->multi) { p_->err_ = "top_rhs: not a multi-GPU handle"; return false; } *d = p_->V.top_rhs; *nd = p_->toprhs_doubles; return true; }
bool Numeric::solve_top_and_bwd(double* drhs) { return p_->solve_top_and_bwd(drhs); }
bool Numeric::set_scaling(int mode, const double* user) { return p_->set_scaling(mode, user); }
bool Numeric::get_scaling(double* out) { return p_->get_scaling(out); }
void Numeric::invalidate_matching() { p_->invalidate_matching(); }
bool Numeric::zero_pivots(std::vector<int>& out) { return p_->zero_pivots(out); }
bool Numeric::failed_pivots(std::vector<int>& out) { return p_->failed_pivots(out); }
bool Numeric::restructure(const Symbolic& S) { return p_->restructure(S); }
bool Numeric::assembly_define(int nseg, const int64_t* off, const int64_t* len) { return p_->assembly_define(nseg, off, len); }
double* Numeric::assembly_buffer(int seg) { return p_->assembly_buffer(seg); }
bool Numeric::assembly_upload(int seg) { return p_->assembly_upload(seg); }
bool Numeric::factor_assembled(const double* scale, const double* shift, FactorStats& st) { return p_->factor_assembled(scale, shift, st); }
bool Numeric::pd_define(const int* dims, const int* ixl, const int* ixu, const int* isl, const int* isu, const int* irn, const int* jcn, const int* segs, int nsegs) { return p_->pd_define(dims, ixl, ixu, isl, isu, irn, jcn, segs, nsegs); }
bool Numeric::pd_put_data(const double* const* arr) { return p_->pd_put_data(arr); }
bool Numeric::pd_put(int vec, const double* const* blocks) { return p_->pd_put(vec, blocks); }
bool Numeric::pd_get(int vec, double* const* blocks) { return p_->pd_get(vec, blocks); }
bool Numeric::pd_solve_once(int rhs, int res, double alpha, double beta) { return p_->pd_solve_once(rhs, res, alpha, beta); }
bool Numeric::pd_residual(int rhs, int res, int resid, const double* deltas, double* norms) { return p_->pd_residual(rhs, res, resid, deltas, norms); }
bool Numeric::ruiz_triplet(int device, int n, int nnz, const int* irn, const int* jcn, const double* a, int base, int sweeps, double* out, std::string& err)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { err = "no HIP device available (no CPU fallback)"; return false; }
    int dev = device; if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
    DeviceGuard guard(dev);
#define RCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); ok = false; break; } } while (0)
    int *di = nullptr, *dj = nullptr; double *da = nullptr, *ds = nullptr; unsigned long long* dm = nullptr;
    bool ok = true;
    do {
        RCHK(hipMalloc((void**)&di, std::max(nnz, 1) * sizeof(int))); RCHK(hipMalloc((void**)&dj, std::max(nnz, 1) * sizeof(int)));
        RCHK(hipMalloc((void**)&da, std::max(nnz, 1) * sizeof(double))); RCHK(hipMalloc((void**)&ds, std::max(n, 1) * sizeof(double)));
        RCHK(hipMalloc((void**)&dm, std::max(n, 1) * sizeof(unsigned long long)));
        RCHK(hipMemcpy(di, irn, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice)); RCHK(hipMemcpy(dj, jcn, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice));
        RCHK(hipMemcpy(da, a, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice));
        RCHK(hipMemset(dm, 0, std::max(n, 1) * sizeof(unsigned long long)));
        const int g1 = std::max(1, std::min(2048, (n + 255) / 256)), g2 = std::max(1, std::min(2048, (nnz + 255) / 256));
        hipLaunchKernelGGL(k_fill, dim3(g1), dim3(256), 0, 0, ds, 1.0, (long long)n);
        for (int it = 0; it < sweeps; ++it) {
            hipLaunchKernelGGL(k_trip_rowmax, dim3(g2), dim3(256), 0, 0, nnz, (const int*)di, (const int*)dj, (const double*)da, (const double*)ds, dm, base);
            hipLaunchKernelGGL(k_trip_rescale, dim3(g1), dim3(256), 0, 0, n, ds, dm);
        }
        RCHK(hipGetLastError());
        RCHK(hipMemcpy(out, ds, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    } while (false);
#undef RCHK
    (void)hipFree(di); (void)hipFree(dj); (void)hipFree(da); (void)hipFree(ds); (void)hipFree(dm);
    return ok;
}
bool Numeric::set_comm_rccl(const void* unique_id128) { return p_->set_comm_rccl(unique_id128); }
bool Numeric::set_comm_callback(int (*fn)(void*, void*, int64_t, int, void*), void* ctx) { return p_->set_comm_callback(fn, ctx); }
bool Numeric::set_comm_range_callback(int (*fn)(void*, void*, int64_t, int, void*, int, int)) { return p_->set_comm_range_callback(fn); }
void Numeric::comm_plan(const Symbolic& S, int nranks, int rank, bool range_local, std::vector<int>& out6) { NumericImpl::comm_plan(S, nranks, rank, range_local, S.n, out6); }
void Numeric::comm_info(int* kind, int* ranks_seen, int* range_local, int* exchange_steps) const
{
    int seen = p_->comm_kind == 0 ? 0 : p_->opt.nranks;
    if (p_->comm_kind == 2 && p_->rccl.comm && p_->rccl.CommCount) { int c = 0; if (p_->rccl.CommCount(p_->rccl.comm, &c) == ncclSuccess) seen = c; }
    if (kind) *kind = p_->comm_kind;
    if (ranks_seen) *ranks_seen = seen;
    if (range_local) *range_local = p_->range_local() ? 1 : 0;
    if (exchange_steps) *exchange_steps = p_->ndepth;
}
long long Numeric::exchange_bytes(int what) const { long long b = 0; for (const auto& sg : p_->rsegs) b += 8 * (what == 0 ? sg.aend - sg.abeg : sg.tend - sg.tbeg); return b; }
bool Numeric::rccl_unique_id(void* out128, std::string& err)
{
    NumericImpl::Rccl R;
    if (!NumericImpl::rccl_load(R, err)) return false;
    ncclUniqueId id;
    ncclResult_t r = R.GetUniqueId(&id);
    if (r != ncclSuccess) { err = std::string("ncclGetUniqueId: ") + (R.GetErrorString ? R.GetErrorString(r) : "error"); return false; }
    std::memcpy(out128, &id, sizeof(id));
    return true;
}

} // namespace mi355x

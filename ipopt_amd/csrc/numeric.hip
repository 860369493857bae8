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
#include "matching_scaling.h"

namespace mi355x {

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    err_ = std::string(#call) + ": " + hipGetErrorString(e_); return false; } } while (0)

// kernel kinds for the profiling entry point (order = include/mi355x_kkt.h MI355X_KKT_KERNEL_*)
enum KernelKind { KK_GATHER_SCALE = 0, KK_FRONT_WAVE, KK_FRONT_LDS64, KK_FRONT_LDS128, KK_BIG_ASSEMBLE, KK_BIG_DIAG, KK_BIG_TRSM,
                  KK_BIG_SCHUR, KK_STATS, KK_SOLVE_PERM, KK_FWD_WAVE, KK_FWD_LDS, KK_FWD_BIG, KK_BWD_WAVE, KK_BWD_LDS, KK_BWD_BIG, KK_FWD_BIG_UPD, KK_BWD_BIG_DOT, KK_COUNT };
#define DBGSTAMP(slot) do { if (V.dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { V.dbg[2 * (slot)] = clock64(); V.dbg[2 * (slot) + 1] = wall_clock64(); } } while (0)
#define DBGT(i) do { if (V.dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) V.dbg[16 + (i)] = clock64(); } while (0)
#define LAUNCH(kind, ...) do { prof_begin(kind); hipLaunchKernelGGL(__VA_ARGS__); prof_end(); } while (0)

static constexpr double BK_ALPHA = 0.6403882032022076;   // (1+sqrt(17))/8
static constexpr double BK_ALPHA0 = 0.1;                 // a diagonal within this factor of its whole remaining column is taken as it comes (no partner search)
static constexpr double PIV_PERT = 1e-10;                // replacement magnitude for a zero pivot
static constexpr int ISG_STRIDE = 8 * 272;               // doubles per big front in DevView::isg
static constexpr double ZERO_REL = 1e-14;                // zero-pivot test relative to the largest entry assembled into the pivot's column

// ------------------------------------------------------------------------------------------------
// device-side view of the symbolic structure + numeric storage (passed by value to kernels)
// ------------------------------------------------------------------------------------------------
// per-front / per-child records in LAUNCH order: one 64-byte load replaces a chain of 4-5 dependent index loads at the
// head of every front kernel (each of them an HBM/MALL round trip on the critical path of a tree level)
struct FrontMeta { int s, c0, k, r0, m, aq0, aq1, ch0, ch1, alias; long long panel_off, cb_off, minv_off; int ldp, ldt;
                   long long cv, wb, gpart; int gbase, gpos, grem, gcols, split, ttab, ttab2, solo, selfasm, bigidx; };      // bigidx: the front's slot in the per-big-front arrays (isg)
struct ChildMeta { int ch, mc, relbase, owner; long long cb_off; int ldt, aliased; long long cvbase, inv; };
// one link of a chain group as seen from a later link of the same group (trailing update, fused solves)
struct GroupLink { long long panel_off, wb, minv_off, cv, tr; int c0, k, m, ldp, r0, ch0, ch1, alias; long long t_off; int ldt, s, selfasm, aq0, aq1, bigidx; };     // t_off/ldt: the link's trailing block (V.cb + t_off)

// Sync-free triangular solves along pure in-place separator chains (a run of consecutive tree levels whose fronts are all chain
// links): ONE launch per sweep for the whole run instead of 1 (forward) / 2 (backward) launches per level.  One workgroup per link
// (+ one per 64 rows beyond the chain in the forward sweep); a link's workgroup waits on a flag for each earlier (forward) / later
// (backward) link, applies that link's 64 x 64 block of the panel to its own rows, then solves with its pivot block and raises its
// own flag -- the point-to-point pipeline of a "synchronisation-free" sparse triangular solve (Liu et al., Euro-Par 2016).
struct ChainLink { long long panel_off, minv_off; int c0, k, ldp, s, r0, koff, fi, pad1; };      // links of all chains, chain by chain, bottom link first; fi: slot of the link's flags
constexpr int FLAG_STRIDE = 32;      // ints between two flags of the sweeps: one 128-byte line each (hundreds of workgroups poll them; side by side they would all queue at one L2 channel)
struct ChainDesc { long long cvb; int link0, nlinks, tail, ktot, wg0f, wg0b;       // cvb: chain vector base, ktot: columns of the chain, wg0*: first workgroup (within the segment's launch)
                   int ch0, ch1, alias0, s0, init, gw0, gw1, tf0, pw0, pw1, dot0, pad0; };         // first link: children (cmeta range), in place on a child's vector, supernode; init: see setup;
                                                                                       // gw0..gw1: tail flags (chwait) awaited before the first link's children are gathered; tf0: own tail flags;
                                                                                       // pw0..pw1: backward, link flags (chwait) of the parent's chain; dot0: first dot workgroup (flags, partial sums)

typedef double v2d __attribute__((ext_vector_type(2)));      // {value, tag}: the 16-byte messages of the solve sweeps
struct DevView {
    // symbolic
    const int* sn_colptr; const int* sn_rowptr; const int* sn_rows; const int* rel;
    const int* child_ptr; const int* child_idx; const int* sn_owner; const int* sn_parent; const int* col_owner;
    const long long* panel_off; const long long* cb_off; const long long* minv_off;
    const int* acolptr; const int* apos; const int* arow; const int* acol;
    const int* dup_ptr; const int* dup_src;
    const int* rslot_ptr; const int* rslot_idx; const int* rslot_col; int rslot_len;
    const int* level_sn;
    const FrontMeta* fmeta;   // parallel to level_sn
    const int* relinv;        // per child of a BIG parent: parent front row -> index in the child's update rows, or -1 (ChildMeta::inv)
    const GroupLink* gtab;    // links of the chain groups (FrontMeta::gbase .. gbase + gpos)
    const int* tile_tab;      // XCD-aware tile orders of the large trailing updates ((ti << 16) | tc), see k_big_schur
    const ChildMeta* cmeta;   // parallel to child_idx
    const int* perm;
    // numeric
    const double* tvals;    // triplet values (device copy)
    double* aval;           // summed + scaled values, permuted lower CSC order
    double* scale;          // symmetric scaling, permuted numbering
    double* scale2;         // second buffer (Jacobi-style equilibration sweeps)
    double* arv;            // |values| in symmetric row-view order (equilibration sweeps stream it)
    unsigned long long* rowmax;  // scratch for equilibration (bit pattern of non-negative doubles)
    double* L;              // panels
    double* cb;             // contribution blocks
    double* wbuf;           // W = L*D copies of the big fronts of the level in flight
    double* minv;           // k x k inverses of the unit-lower pivot blocks (column-major, ld = k)
    double* dinv; double* doff; int* ptype; int* lperm;
    int4*   fstat;          // per front {neg, zero, two, small}
    double* xw;             // work vector (permuted, scaled)
    double* cvec;           // forward-solve contributions, aligned with sn_rows
    double* bw; double* xacc;   // iterative refinement: scaled right-hand side and accumulated solution (permuted numbering)
    double* gpart;          // partial sums of the backward dot products of the chain groups
    double* zb;             // z = D^{-1} y of the forward sweep (pivot order); xw keeps b until the backward sweep writes x
    double* ybuf;           // y of the pivot rows (big fronts: the update rows are handled by a second, multi-workgroup launch)
    // multi-GPU top arena (full m x m squares per replicated front), null on 1 GPU
    double* arena; const long long* arena_off;
    double* top_rhs; const long long* top_rhs_off;
    // parameters
    double pivtol, pivtol2, small;   // u, the largest u IncreaseQuality may reach (decision-change tracking), absolute zero threshold
    int* colfail;           // per column of a BIG front: 1 once some multiplier of L21 exceeded 1/u (a posteriori test, k_big_trsm)
    int* qstat;             // [0]: some pivot decision of this factorisation would differ at u = pivtol2
    double* cnorm;          // inf-norm of every column of the (scaled) INPUT matrix, permuted numbering: scale of the zero-pivot test
    const ChainLink* chlink; const ChainDesc* chdesc;     // chain solve tables
    int strace_b;                    // (first backward workgroup's slot)
    unsigned long long* strace;      // development aid (MI355X_KKT_SOLVE_TRACE=file): 4 wall-clock stamps per workgroup of the data-flow sweeps
    const int* chwg_f; const int* chwg_b; const int* chwait; int* sflag_t; int* sflag_dot; double* dpart; v2d* ytag; v2d* xtag;     // workgroup -> chain (forward / backward launch), wait lists, tail / gather flags
    int* tcnt;                                                  // per front: panel-solve workgroups finished (fused pivot block + panel solve + narrow update launch), zeroed by the prologue
    double* isg; int* hasis;      // per big front: the four 16 x 16 diagonal-block inverses of L11 left by the blocked factorisation (hasis: valid), for the panel solves
    int* sflag_s;           // [4 * link + q]: rows of the group's link q have stored their W / L against this link (k_grp_fused)
    int* sflag_b; int* sflag_d; int* sepoch;     // (sflag_d / sepoch[2]: pivot block done, fused pivot-block + panel-solve launch)              // per-supernode 'done' flags of the chain sweeps (value = epoch of the solve)
    int* zpiv;              // per column (permuted numbering): 1 if its pivot was a zero pivot (DetermineDependentRows)
    int n, nnz_a, nsn, rank;
    int fastpiv;            // pivot blocks of the big fronts: blocked LDL^T accepted a posteriori first, the strict loop as fall-back (ldlt_blocked_static)
    double fastu;           // ... accepted iff every multiplier <= 1 / max(u, u2, fastu)
    int asm_pull;           // k_big_assemble: entries summed through the inverse row maps and written once (default) / scatter-added child by child
    unsigned long long* dbg;   // optional phase time stamps of block 0 (development aid), may be null
};

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
// ---- wavefront reductions on the DPP path (row-local butterflies, then 4 readlanes): ~10x lower latency than the
// ds_bpermute shuffles on the serial pivot chain.  Results are wave-uniform. ----
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double x, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
__device__ __forceinline__ double wave_max_all(double x)
{
    x = fmax(x, dpp_f64<0xB1>(x));    // quad_perm [1,0,3,2]
    x = fmax(x, dpp_f64<0x4E>(x));    // quad_perm [2,3,0,1]
    x = fmax(x, dpp_f64<0x141>(x));   // row_half_mirror
    x = fmax(x, dpp_f64<0x140>(x));   // row_mirror  -> every lane of a 16-lane row holds the row maximum
    return fmax(fmax(readlane_f64(x, 0), readlane_f64(x, 16)), fmax(readlane_f64(x, 32), readlane_f64(x, 48)));
}
__device__ __forceinline__ unsigned long long wave_or_all(unsigned long long v)
{
    int lo = (int)(v & 0xffffffffull), hi = (int)(v >> 32);
    lo |= __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false);  hi |= __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false);
    lo |= __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xF, 0xF, false);  hi |= __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xF, 0xF, false);
    lo |= __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xF, 0xF, false); hi |= __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xF, 0xF, false);
    lo |= __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xF, 0xF, false); hi |= __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xF, 0xF, false);
    const unsigned int l = (unsigned)(__builtin_amdgcn_readlane(lo, 0) | __builtin_amdgcn_readlane(lo, 16) | __builtin_amdgcn_readlane(lo, 32) | __builtin_amdgcn_readlane(lo, 48));
    const unsigned int h = (unsigned)(__builtin_amdgcn_readlane(hi, 0) | __builtin_amdgcn_readlane(hi, 16) | __builtin_amdgcn_readlane(hi, 32) | __builtin_amdgcn_readlane(hi, 48));
    return ((unsigned long long)h << 32) | l;
}
// sum over the 64 lanes on the DPP path (row-local butterflies + 4 readlanes), wave-uniform result
__device__ __forceinline__ double wave_sum_dpp(double x)
{
    x += dpp_f64<0xB1>(x);
    x += dpp_f64<0x4E>(x);
    x += dpp_f64<0x141>(x);
    x += dpp_f64<0x140>(x);
    return (readlane_f64(x, 0) + readlane_f64(x, 16)) + (readlane_f64(x, 32) + readlane_f64(x, 48));
}
__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
    return x;
}

// ------------------------------------------------------------------------------------------------
// value gather (duplicates summed in a fixed order => bitwise reproducible) and equilibration
// ------------------------------------------------------------------------------------------------
__global__ void k_gather_values(DevView V)
{
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < V.nnz_a; q += gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int p = V.dup_ptr[q]; p < V.dup_ptr[q + 1]; ++p) s += V.tvals[V.dup_src[p]];
        V.aval[q] = s;
    }
}
// Device-side KKT value assembly (replaces TripletHelper::FillValues over the whole CompoundSymMatrix,
// IpTripletHelper.cpp:249-362, and the 8 nnz-byte PCIe copy): the triplet value array is a concatenation of SEGMENTS, each
//   tvals[off + i] = scale * src[i] + shift
// with a device-resident source (W, J_c, J_d values; the Sigma / D diagonals) and two scalars per segment: W_factor,
// delta_x/s, -delta_c/d, the -1 of the (4,2) identity block.  A retry that only changes the deltas uploads nothing.
constexpr int ASM_MAXSEG = 16;
struct AsmSegs { int nseg; long long off[ASM_MAXSEG], len[ASM_MAXSEG]; const double* src[ASM_MAXSEG]; double scale[ASM_MAXSEG], shift[ASM_MAXSEG]; };
__global__ void k_assemble_segments(double* tvals, AsmSegs A)
{
    const int sgi = blockIdx.y;
    if (sgi >= A.nseg) return;
    const long long len = A.len[sgi];
    const double sc = A.scale[sgi], sh = A.shift[sgi];
    const double* src = A.src[sgi];
    double* dst = tvals + A.off[sgi];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (long long)gridDim.x * blockDim.x)
        dst[i] = (sc != 0.0 ? sc * src[i] : 0.0) + sh;
}
__global__ void k_fill(double* p, double v, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_zero_u64(unsigned long long* p, int n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0ull;
}
// Ruiz equilibration as gathers over the symmetric row view of the pattern (no atomics, Jacobi style => deterministic):
//   k_abs_rowview   arv(p) = |a(slot(p))|, the one pass with scattered reads; every sweep then streams arv
//   k_ruiz_sweep    s_new(i) = s(i) / sqrt( s(i) max_p arv(p) s(col(p)) ), 8 lanes per row, shuffle max; sin == null means 1
__global__ void k_abs_rowview(DevView V)
{
    const int total = V.rslot_len;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < total; p += gridDim.x * blockDim.x) V.arv[p] = fabs(V.aval[V.rslot_idx[p]]);
}
__global__ void k_ruiz_sweep(DevView V, const double* sin, double* sout, double* cnorm_out)
{
    const int sub = threadIdx.x & 7;
    for (int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 3; i < ((V.n + 7) & ~7) + 0; i += (gridDim.x * blockDim.x) >> 3) {
        double mx = 0.0;
        if (i < V.n) {
            const int p1 = V.rslot_ptr[i + 1];
            for (int p = V.rslot_ptr[i] + sub; p < p1; p += 8) mx = fmax(mx, V.arv[p] * (sin ? sin[V.rslot_col[p]] : 1.0));
        }
        mx = fmax(mx, __shfl_xor(mx, 1)); mx = fmax(mx, __shfl_xor(mx, 2)); mx = fmax(mx, __shfl_xor(mx, 4));
        if (i < V.n && sub == 0) {
            const double si = sin ? sin[i] : 1.0; mx *= si; sout[i] = mx > 0.0 ? si / sqrt(mx) : si;
            if (cnorm_out) cnorm_out[i] = 1.0;        // after the last sweep every row / column of the scaled matrix has inf-norm ~ 1: the scale of the zero-pivot test
        }
    }
}
// user-supplied symmetric scaling (original numbering) -> permuted numbering (the MA97 "reuse the caller-held factors" mode)
__global__ void k_user_scale(DevView V, const double* s_orig)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < V.n; i += gridDim.x * blockDim.x) V.scale[i] = s_orig[V.perm[i]];
}
// stand-alone symmetric Ruiz equilibration of a TRIPLET matrix (for hosts that scale outside the solver: Ipopt's
// TSymScalingMethod hook, IpTSymLinearSolver.cpp:429-441,511-514).  Row maxima by atomicMax on the bit pattern of the
// non-negative doubles (max is order independent => deterministic).
__global__ void k_trip_rowmax(int nnz, const int* irn, const int* jcn, const double* a, const double* s, unsigned long long* rowmax, int base)
{
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nnz; q += gridDim.x * blockDim.x) {
        const int i = irn[q] - base, j = jcn[q] - base;
        const double v = fabs(a[q]) * s[i] * s[j];
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        atomicMax(&rowmax[i], b);
        if (j != i) atomicMax(&rowmax[j], b);
    }
}
__global__ void k_trip_rescale(int n, double* s, unsigned long long* rowmax)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double mx = __longlong_as_double((long long)rowmax[i]);
        if (mx > 0.0) s[i] /= sqrt(mx);
        rowmax[i] = 0ull;
    }
}
// inf-norm of every row (= column) of the scaled input matrix over the symmetric row view: cnorm(i) = s(i) max_p arv(p) s(col(p))
__global__ void k_colnorm(DevView V, int scaled)
{
    const int sub = threadIdx.x & 7;
    for (int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 3; i < ((V.n + 7) & ~7); i += (gridDim.x * blockDim.x) >> 3) {
        double mx = 0.0;
        if (i < V.n) {
            const int p1 = V.rslot_ptr[i + 1];
            for (int p = V.rslot_ptr[i] + sub; p < p1; p += 8) mx = fmax(mx, V.arv[p] * (scaled ? V.scale[V.rslot_col[p]] : 1.0));
        }
        mx = fmax(mx, __shfl_xor(mx, 1)); mx = fmax(mx, __shfl_xor(mx, 2)); mx = fmax(mx, __shfl_xor(mx, 4));
        if (i < V.n && sub == 0) V.cnorm[i] = mx * (scaled ? V.scale[i] : 1.0);
    }
}
__global__ void k_apply_scale(DevView V)
{
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < V.nnz_a; q += gridDim.x * blockDim.x)
        V.aval[q] *= V.scale[V.arow[q]] * V.scale[V.acol[q]];
}

// In-place inverse of the unit lower triangular k x k block at the top of F by recursive doubling:
//   [A 0; B C]^{-1} = [A^{-1} 0; -C^{-1} B A^{-1}  C^{-1}],  block size h = 1, 2, 4, ...
// Each stage is two fully parallel small products (T = B A^{-1} parked in the unused mirror position above the
// diagonal, then B <- -C^{-1} T): log2(k) stages of 2 barriers instead of a k-step substitution chain.  Every solve
// then multiplies by L11^{-1}.
typedef double v4f64_ __attribute__((ext_vector_type(4)));
template <int NT>
__device__ __forceinline__ void invert_unit_lower(double* F, const int ld, const int k)
{
    const int tid = threadIdx.x;
    for (int sh = 0; (1 << sh) < k; ++sh) {
        const int h = 1 << sh;
        const int npair = (k + 2 * h - 1) >> (sh + 1);
        if (h >= 16) {
            // the last stages carry ~90 % of the n^3/3 flops: 16 x 16 output tiles on v_mfma_f64_16x16x4_f64 (one wavefront per
            // tile, operands straight from LDS with the triangular / unit-diagonal masks applied on the fly).
            //   D[l4 + 4g][l15] = sum_k A[l15][k = l4] B[k = l4][l15]
            const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
            const int tpp = (h >> 4) * (h >> 4), ntile = npair * tpp;
            for (int tile = wave; tile < ntile; tile += NT / 64) {          // phase 1: T = B A^{-1}  -> mirror position
                const int pr = tile / tpp, rem = tile - pr * tpp, ib = rem / (h >> 4), cb = rem - ib * (h >> 4);
                const int o = pr << (sh + 1);
                if (o + h >= k) continue;                                   // no B block in this pair
                const int gi = o + h + ib * 16 + l15;                       // A operand row (a row of B)
                const int c = cb * 16 + l15;                                // B operand column (a column of A^{-1})
                v4f64_ acc = (v4f64_){0.0, 0.0, 0.0, 0.0};
                for (int p0 = cb * 16; p0 < h; p0 += 4) {
                    const int pp = p0 + l4;
                    const double av = (gi < k) ? F[gi + (o + pp) * ld] : 0.0;
                    const double bv = (pp > c) ? F[o + pp + (o + c) * ld] : (pp == c ? 1.0 : 0.0);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) { const int ri = o + h + ib * 16 + l4 + 4 * g; if (ri < k) F[(o + c) + ri * ld] = acc[g]; }
            }
            __syncthreads();
            for (int tile = wave; tile < ntile; tile += NT / 64) {          // phase 2: B <- -C^{-1} T
                const int pr = tile / tpp, rem = tile - pr * tpp, ib = rem / (h >> 4), cb = rem - ib * (h >> 4);
                const int o = pr << (sh + 1);
                if (o + h >= k) continue;
                const int i = ib * 16 + l15;                                // A operand row (a row of C^{-1})
                const int c = cb * 16 + l15;                                // B operand column (a column of T)
                v4f64_ acc = (v4f64_){0.0, 0.0, 0.0, 0.0};
                for (int p0 = 0; p0 < (ib + 1) * 16; p0 += 4) {
                    const int pp = p0 + l4;
                    const double av = (i > pp) ? ((o + h + i < k) ? F[o + h + i + (o + h + pp) * ld] : 0.0) : (i == pp ? 1.0 : 0.0);
                    const double bv = (o + h + pp < k) ? F[(o + c) + (o + h + pp) * ld] : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) { const int ri = o + h + ib * 16 + l4 + 4 * g; if (ri < k) F[ri + (o + c) * ld] = -acc[g]; }
            }
            __syncthreads();
            continue;
        }
        const int total = npair << (2 * sh);
        for (int e = tid; e < total; e += NT) {
            const int pr = e >> (2 * sh), rem = e & ((1 << (2 * sh)) - 1);
            const int i = rem & (h - 1), c = rem >> sh, o = pr << (sh + 1);
            const int gi = o + h + i, gc = o + c;
            if (gi >= k) continue;
            double acc = F[gi + gc * ld];                                   // p = c term (A^{-1}(c,c) = 1)
            for (int p = c + 1; p < h; ++p) acc += F[gi + (o + p) * ld] * F[o + p + gc * ld];
            F[gc + gi * ld] = acc;                                          // T(i,c) -> mirror position (upper part)
        }
        __syncthreads();
        for (int e = tid; e < total; e += NT) {
            const int pr = e >> (2 * sh), rem = e & ((1 << (2 * sh)) - 1);
            const int i = rem & (h - 1), c = rem >> sh, o = pr << (sh + 1);
            const int gi = o + h + i, gc = o + c;
            if (gi >= k) continue;
            double acc = F[gc + gi * ld];                                   // p = i term (C^{-1}(i,i) = 1)
            for (int p = 0; p < i; ++p) acc += F[gi + (o + h + p) * ld] * F[gc + (o + h + p) * ld];
            F[gi + gc * ld] = -acc;
        }
        __syncthreads();
    }
}

// ================================================================================================
// Register-tiled LDL^T core (the production path).  The assembled front is pulled from LDS into VGPRs as a
// G x G grid of TS x TS tiles (full symmetric storage, thread (ti,tj) owns rows ti*TS.., columns tj*TS..);
// LDS only carries the pivot column(s) of the current step (published by the G owner threads, read by everybody:
// by symmetry the same vector serves as row and column multipliers) and the finished L columns.
//   * no interchanges: Bunch-Kaufman picks the pivot among the still-alive fully-summed rows and the chosen
//     PHYSICAL row is eliminated in place; the pivot order `ord` is applied once at write-back;
//   * one barrier per 1x1 pivot (two when the BK test needs the partner column), none of them inside a wavefront's
//     own dependency chain when the workgroup is a single wave (fronts of order <= 64);
//   * pivot search: DPP wave reduction, done redundantly by every wave on the published column.
// (NT,TS) = (64,4): order <= 32, (64,8): <= 64, (256,8): <= 128, (256,4): the 64-column pivot block of a big front.
// ================================================================================================
template <int TS>
__device__ __forceinline__ void publish_col(double* buf, const double (&t)[TS][TS], const int row0, const int jl)
{
    // jl is wave-uniform: a scalar branch selects the STATICALLY indexed register column.  The empty asm keeps the
    // cases apart -- merged, they become a dynamically indexed t[a][jl] and the whole tile array moves to scratch.
#pragma unroll
    for (int b = 0; b < TS; ++b)
        if (jl == b) {
#pragma unroll
            for (int a = 0; a < TS; ++a) { double v = t[a][b]; asm volatile("" : "+v"(v)); buf[row0 + a] = v; }
        }
}

#ifdef MI355X_PIVSTAT
__device__ unsigned long long g_fstat[32];
__device__ unsigned long long g_dt[16], g_dtacc[16];     // fused pivot block + panel solve, single front: wall-clock stamps of one launch / sums over launches
__device__ unsigned long long g_pivstat[16];      // development build only: [0] pivots steps, [1] slow-path entries, [2] quick accepts, [3] exact path, [4] pass-overs, [5] 2x2, [6] no-partner
#ifdef MI355X_PIVSTAT_COUNT
#define PIVSTAT(i) do { if (threadIdx.x == 0) atomicAdd(&g_pivstat[i], 1ull); } while (0)
#else
#define PIVSTAT(i) do { } while (0)
#endif
#else
#define PIVSTAT(i) do { } while (0)
#endif

__device__ __forceinline__ double fast_rcp(double d)
{
    double r = __builtin_amdgcn_rcp(d);          // v_rcp_f64 + two Newton steps: full fp64 accuracy without the division macro
    double e = fma(-d, r, 1.0); r = fma(r, e, r);
    e = fma(-d, r, 1.0); r = fma(r, e, r);
    return r;
}

// Threshold pivoting (what u = pivtol means here; DESIGN.md "pivoting"):
//   * candidate order: the still-alive fully-summed rows in physical order; a candidate j whose diagonal is within
//     alpha0 = 0.1 of its whole remaining column (|a_jj| >= alpha0 max_{i != j} |a_ij|: the threshold test of UMFPACK / MA48
//     at their default u, seven orders of magnitude tighter than Ipopt's 1e-8) is eliminated as a 1x1 without looking
//     further; otherwise the Bunch-Kaufman rule (alpha = 0.64, on the alive fully-summed part) PREFERS one of
//     {1x1 at j, 1x1 at r, 2x2 (j,r)}, r = the fully-summed row with the largest |a_rj|;
//   * a pivot is ACCEPTED only if it passes the MA27/MA57 threshold tests against the WHOLE remaining front column --
//     alive fully-summed rows AND update rows:   1x1: |a_pp| >= u max_{i != p} |a_ip|;
//     2x2: |E^{-1}| (g_p, g_q)^T <= 1/u componentwise, g = column maxima outside the block (Duff & Reid 1983; MA57);
//     if the preferred pivot fails, the other two are tried;
//   * a candidate with no acceptable pivot is PASSED OVER (retried after the next elimination has updated it: the
//     in-front part of MA27's delayed pivoting).  When every alive candidate has failed, the structure being static
//     (no delay to the parent front), the first one is eliminated anyway by the plain Bunch-Kaufman choice and counted in
//     `ndelay` (reported as num_delay; the reference adapters read the same counter from MA97/SPRAL);
//   * zero test relative to what was assembled into the candidate's own column: a candidate whose whole remaining column is
//     <= max(small, 1e-14 max_i |F(i,j)| at assembly) is a zero pivot => SYMSOLVER_SINGULAR;
//   * `chg` is set when some decision would come out differently at u2 (= pivtolmax): IncreaseQuality uses it to
//     answer "can a larger u change the factorisation at all".
// Big fronts: the pivot block only sees its k x k block here (ext rows are checked a posteriori in k_big_trsm).
template <int NT, int TS, bool WIDE>
__device__ __forceinline__ void ldlt_reg(double (&t)[TS][TS], const int m, const int k, double* Lbuf, const int ldL, double* colbuf,
                                         double* dinv_s, double* doff_s, int* pt_s, int* ord, const double u, const double u2, const double small, const double* cm0, const double cmx, int* zp,
                                         int& nneg, int& nzero, int& ntwo, int& ndelay, int& chg)
{
    // The per-pivot instruction stream IS the critical path (measured: ~5 cycles per wave instruction), so the common
    // case -- 1x1 pivot on the first alive row -- is kept to ~100 instructions: unconditional wide LDS reads + bit-mask
    // selects, ballots instead of max-reductions for the Bunch-Kaufman and threshold acceptance tests, reciprocal by
    // v_rcp_f64 + Newton, L column written by the 16 (8) threads that already hold it.
    constexpr int G = (NT == 64) ? 8 : (NT == 1024 ? 32 : 16);
    constexpr int MAXM = G * TS;
    constexpr bool TWO = MAXM > 64;              // each lane looks at rows `lane` and `lane + 64` of a published column
    const int tid = threadIdx.x, lane = tid & 63;
    const int ti = tid % G, tj = tid / G;
    const int row0 = ti * TS, col0 = tj * TS;
    unsigned rowvalid = 0;
#pragma unroll
    for (int x = 0; x < TS; ++x) if (row0 + x < m) rowvalid |= 1u << x;
    const unsigned long long lanebit = 1ull << lane;
    const int lane1 = TWO ? min(lane + 64, MAXM - 1) : 0;
    const bool up0 = lane >= k && lane < m;                   // update rows seen by this lane
    const bool up1 = TWO && lane + 64 >= k && lane + 64 < m;
    // fully-summed rows not yet eliminated: rows 0..63 in alive, rows 64..127 in alive1 (WIDE: pivot blocks of <= 128 columns)
    unsigned long long alive = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
    unsigned long long alive1 = (WIDE && k > 64) ? ((k >= 128) ? ~0ull : ((1ull << (k - 64)) - 1ull)) : 0ull;
    unsigned long long tryb = alive, tryb1 = alive1;          // candidates not yet passed over since the last elimination
    bool force = false;
    double u2e = u2;                                          // 0 while forced: a forced pivot stays forced at any larger u
    unsigned long long chgm = 0ull;
    const double zmax = fmax(small, ZERO_REL * cmx);
    auto clear_row = [&](int r) { if (!WIDE || r < 64) alive &= ~(1ull << r); else alive1 &= ~(1ull << (r - 64)); };
    int step = 0, bufsel = 0;
#ifdef MI355X_PIVSTAT
    long long tprev = clock64(); bool was_slow = false; int first = 1; unsigned long long acc_f = 0, acc_s = 0, n_f = 0, n_s = 0;
#endif
    while ((alive | alive1) != 0ull) {
#ifdef MI355X_PIVSTAT
        if (NT == 256 && TS == 4 && tid == 0 && gridDim.x == 1 && blockIdx.y == 0) { const long long tn = clock64(); if (!first) { if (was_slow) { acc_s += tn - tprev; n_s++; } else { acc_f += tn - tprev; n_f++; } } tprev = tn; first = 0; was_slow = false; }
#endif
        double* colA = colbuf + bufsel * 2 * MAXM; bufsel ^= 1;
        double* colB = colA + MAXM;
        const int j = __builtin_amdgcn_readfirstlane(tryb != 0ull ? __ffsll((long long)tryb) - 1 : 64 + __ffsll((long long)tryb1) - 1);
        if (tj == j / TS) publish_col<TS>(colA, t, row0, j % TS);
        __syncthreads();
        // every LDS read of the common case (1x1 pivot on row j) is issued here, in ONE round trip
        const double djj = colA[j];
        const double avr = colA[lane];
        const double avr1 = TWO ? colA[lane1] : 0.0;
        double rv[TS], cv[TS];
#pragma unroll
        for (int x = 0; x < TS; ++x) { rv[x] = colA[row0 + x]; cv[x] = colA[col0 + x]; }
        const double ajj = fabs(djj);
        const double f0 = fabs(avr), f1 = fabs(avr1);
        const bool cand = (alive & lanebit) != 0ull && lane != j;
        const bool cand1 = WIDE && (alive1 & lanebit) != 0ull && lane + 64 != j;
        const double av0 = cand ? f0 : -1.0, av1 = cand1 ? f1 : -1.0;
        const double av = fmax(av0, av1);                      // alive fully-summed rows (the Bunch-Kaufman candidates)
        const double ga = fmax(fmax(av, up0 ? f0 : 0.0), up1 ? f1 : 0.0);      // whole remaining column, diagonal excluded
        // the common case must stay ONE straight instruction stream (every taken branch costs an instruction refetch on the
        // serial pivot chain): three ballots OR-ed into one scalar test, statistics accumulated branch-free
        const unsigned long long slowm = __ballot(ga * BK_ALPHA0 > ajj)       // some |a_ij| > |a_jj| / alpha0: full Bunch-Kaufman test
                                       | __ballot(ga * u > ajj)               // 1x1 at j fails the threshold test
                                       | __ballot(!(ajj > zmax));             // possibly a (numerically) zero diagonal: exact test below
        chgm |= __ballot(ga * u2e > ajj);
        double d = djj;                    // 1x1 pivot value on physical row p (pivot column in rv / cv)
        int p = j;
        PIVSTAT(0);
        if (__builtin_expect(slowm != 0ull, 0)) {
            PIVSTAT(1);
#ifdef MI355X_PIVSTAT
            was_slow = true;
#endif
            const int bi = (av1 > av0) ? lane + 64 : lane;       // this lane's best candidate row
            const double ztol = fmax(small, ZERO_REL * cm0[j]);     // zero threshold of candidate j
            const double uu = force ? 0.0 : u;
            const double lam = wave_max_all(av);               // -1: no alive fully-summed partner
            int sel = -1;                                      // 0: 1x1 at j, 1: 1x1 at r, 2: 2x2 (j, r)
            int r = -1;
            bool zero = false;
            if (lam > 0.0) {
#ifdef MI355X_PIVSTAT_COUNT
                { const double rho = ajj / lam; if (threadIdx.x == 0) atomicAdd(&g_pivstat[rho >= 0.5 ? 12 : (rho >= 0.25 ? 13 : (rho >= 0.1 ? 14 : 15))], 1ull); if (threadIdx.x == 0 && rho < 0.01) atomicAdd(&g_pivstat[7], 1ull); }
#endif
                const unsigned long long hit = __ballot(av == lam);
                const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)hit) - 1);
                r = __builtin_amdgcn_readlane(bi, src);
                if (tj == r / TS) publish_col<TS>(colB, t, row0, r % TS);
                __syncthreads();
                const double h0 = fabs(colB[lane]), h1 = TWO ? fabs(colB[lane1]) : 0.0;
                const bool cs = (alive & lanebit) != 0ull && lane != r;
                const bool cs1 = WIDE && (alive1 & lanebit) != 0ull && lane + 64 != r;
                const double sfs = fmax(cs ? h0 : 0.0, cs1 ? h1 : 0.0);
                const double sig = wave_max_all(sfs);                                            // Bunch-Kaufman sigma
                const double hall = fmax(fmax(sfs, up0 ? h0 : 0.0), up1 ? h1 : 0.0);             // whole column r, diagonal excluded
                const double a = djj, b = colA[r], c = colB[r];
                const double arr = fabs(c), ab = fabs(b);
                const double det = a * c - b * b, adet = fabs(det);
                const double ztr = fmax(small, ZERO_REL * cm0[r]);
                const bool nz2 = adet > fmax(small, ZERO_REL * fmax(ajj * arr, ab * ab));     // the block itself is not (numerically) singular
                const int pref = (ajj >= BK_ALPHA * lam || ajj * sig >= BK_ALPHA * lam * lam) ? 0 : ((arr >= BK_ALPHA * sig) ? 1 : 2);
                // ONE reduction bounds every column maximum the threshold tests need (G >= gj, gr, gj2, gr2): when the
                // preferred pivot passes them with G -- at u and at u2, the usual case -- it passes the exact tests too
                const double G = wave_max_all(fmax(ga, hall));
                const double um = fmax(uu, u2e);
                const bool quick = !force && ((pref == 0) ? (ajj > ztol && ajj >= um * G)
                                            : (pref == 1) ? (arr > ztr && arr >= um * G)
                                                          : (nz2 && (arr + ab) * G * um <= adet && (ab + ajj) * G * um <= adet));   // (forced pivots are counted exactly)
                if (quick) { sel = pref; PIVSTAT(2); }
                else {
                    PIVSTAT(3);
                    const double gj = wave_max_all(ga);
                    const double gr = wave_max_all(hall);
                    const bool nj0 = lane != j, nj1 = lane + 64 != j;
                    const double gj2 = wave_max_all(fmax(fmax((cand && lane != r) ? f0 : 0.0, (cand1 && lane + 64 != r) ? f1 : 0.0), fmax(up0 ? f0 : 0.0, up1 ? f1 : 0.0)));
                    const double gr2 = wave_max_all(fmax(fmax((cs && nj0) ? h0 : 0.0, (cs1 && nj1) ? h1 : 0.0), fmax(up0 ? h0 : 0.0, up1 ? h1 : 0.0)));
                    const double t1 = arr * gj2 + ab * gr2, t2 = ab * gj2 + ajj * gr2;                // |E^{-1}| (gj2, gr2)^T |det|
                    const bool ok0 = ajj > ztol && ajj >= uu * gj;
                    const bool ok1 = arr > ztr && arr >= uu * gr;
                    const bool ok2 = nz2 && t1 * uu <= adet && t2 * uu <= adet;
                    if ((pref == 0 && ok0) || (pref == 1 && ok1) || (pref == 2 && ok2)) sel = pref;
                    else if (ok0) sel = 0; else if (ok2) sel = 2; else if (ok1) sel = 1;
                    if (sel >= 0) {
                        const bool f_u  = (sel == 0) ? (ajj < u * gj)  : ((sel == 1) ? (arr < u * gr)  : (t1 * u > adet  || t2 * u > adet));
                        const bool f_u2 = (sel == 0) ? (ajj < u2 * gj) : ((sel == 1) ? (arr < u2 * gr) : (t1 * u2 > adet || t2 * u2 > adet));
                        if (f_u) ndelay += (sel == 2) ? 2 : 1;     // only possible when forced
                        if (f_u2 && !force) chgm |= 1ull;          // (a forced pivot stays forced at any larger u)
                    }
                }
            } else {
                PIVSTAT(6);
                const double gj = wave_max_all(ga);
                if (ajj > ztol && ajj >= uu * gj) { sel = 0; if (ajj < u * gj) ndelay += 1; }
                else if (!(ajj > ztol) && !(gj > ztol)) { sel = 0; zero = true; }          // the whole remaining column is zero
            }
            if (sel < 0) {
                if (!force) {          // pass over: retried once another elimination has updated the column
                    PIVSTAT(4);
                    if (!WIDE || j < 64) tryb &= ~(1ull << j); else tryb1 &= ~(1ull << (j - 64));
                    if ((tryb | tryb1) == 0ull) { force = true; u2e = 0.0; tryb = alive; tryb1 = alive1; }   // every candidate failed: static pivoting
                    continue;
                }
                sel = 0; zero = true;  // forced and nothing usable: a (perturbed) zero pivot => singular
            }
            force = false; u2e = u2;
            if (sel == 2) {
                PIVSTAT(5);
                const int q = r;
                const double a = colA[p], b = colA[q], c = colB[q];
                const double det = a * c - b * b;
                const double idet = fast_rcp(det);
                double l0[TS], l1[TS], w0[TS], w1[TS];
#pragma unroll
                for (int x = 0; x < TS; ++x) {
                    const double r0 = colA[row0 + x], r1 = colB[row0 + x];
                    l0[x] = (c * r0 - b * r1) * idet; l1[x] = (a * r1 - b * r0) * idet;
                    w0[x] = colA[col0 + x]; w1[x] = colB[col0 + x];
                }
#pragma unroll
                for (int x = 0; x < TS; ++x)
#pragma unroll
                    for (int y = 0; y < TS; ++y) t[x][y] -= l0[x] * w0[y] + l1[x] * w1[y];
                if (tj == 0) {
#pragma unroll
                    for (int x = 0; x < TS; ++x) if ((rowvalid >> x) & 1u) { Lbuf[row0 + x + step * ldL] = l0[x]; Lbuf[row0 + x + (step + 1) * ldL] = l1[x]; }
                }
                if (tid == 0) {
                    ord[step] = p; ord[step + 1] = q; pt_s[step] = 2; pt_s[step + 1] = 3;
                    dinv_s[step] = c * idet; dinv_s[step + 1] = a * idet; doff_s[step] = -b * idet; doff_s[step + 1] = 0.0;
                }
                if (det < 0.0) nneg += 1; else if (a + c < 0.0) nneg += 2;
                ntwo++; clear_row(p); clear_row(q); step += 2;
                tryb = alive; tryb1 = alive1;
                continue;
            }
            if (sel == 1) {
                p = r; d = colB[r];
#pragma unroll
                for (int x = 0; x < TS; ++x) { rv[x] = colB[row0 + x]; cv[x] = colB[col0 + x]; }
            }
            if (zero) { nzero++; d = (d < 0.0) ? -PIV_PERT : PIV_PERT; if (tid == 0) zp[p] = 1; }
        }
        {   // 1x1 pivot on physical row p
            const double di = fast_rcp(d);
            // No masking of dead rows / columns: the rank-1 update itself annihilates row and column p (l_p = 1 up to
            // rounding), padding rows are exact zeros, and whatever residue is left in dead positions is never read
            // (alive mask in the search, (i > c) filter at write-back).  That removes ~50 instructions per pivot.
            double l0[TS];
#pragma unroll
            for (int x = 0; x < TS; ++x) l0[x] = rv[x] * di;
            const double (&w0)[TS] = cv;
#pragma unroll
            for (int x = 0; x < TS; ++x)
#pragma unroll
                for (int y = 0; y < TS; ++y) t[x][y] -= l0[x] * w0[y];
            if (tj == 0) {
#pragma unroll
                for (int x = 0; x < TS; ++x) if ((rowvalid >> x) & 1u) Lbuf[row0 + x + step * ldL] = l0[x];
            }
            if (tid == 0) { ord[step] = p; pt_s[step] = 1; dinv_s[step] = di; doff_s[step] = 0.0; }
            nneg += (d < 0.0) ? 1 : 0;
            clear_row(p); step += 1;
            tryb = alive; tryb1 = alive1;
        }
    }
#ifdef MI355X_PIVSTAT
    if (n_f + n_s) { atomicAdd(&g_pivstat[8], acc_f); atomicAdd(&g_pivstat[9], acc_s); atomicAdd(&g_pivstat[10], n_f); atomicAdd(&g_pivstat[11], n_s); }
#endif
    if (chgm != 0ull) chg = 1;
    __syncthreads();
}

// zero-pivot scale of every fully-summed COLUMN -> cm0[0..k): the larger of the column's inf-norm in the (scaled) input matrix
// and of what was assembled into it in this front.  A pivot is numerically zero relative to ITS column -- a huge Sigma entry
// elsewhere in the front (late barrier iterations, no equilibration) must not make healthy small pivots look like zeros, and a
// column whose entries already cancelled in the children (dependent constraint rows) must still be measured against what it was.
template <int NT, int TS>
__device__ __forceinline__ double front_colmax(const double (&t)[TS][TS], double* cm0, const int k, const double* cnorm)
{
    constexpr int G = (NT == 64) ? 8 : (NT == 1024 ? 32 : 16);
    const int tid = threadIdx.x, ti = tid % G, tj = tid / G;
#pragma unroll
    for (int y = 0; y < TS; ++y) {
        double mx = 0.0;
#pragma unroll
        for (int x = 0; x < TS; ++x) mx = fmax(mx, fabs(t[x][y]));
        // the G threads that share tile column tj are G consecutive lanes (tid = ti + G tj): butterfly inside the group
        mx = fmax(mx, dpp_f64<0xB1>(mx));
        mx = fmax(mx, dpp_f64<0x4E>(mx));
        mx = fmax(mx, dpp_f64<0x141>(mx));           // 8 lanes
        if (G >= 16) mx = fmax(mx, dpp_f64<0x140>(mx));   // 16 lanes
        if (G == 32) mx = fmax(mx, __shfl_xor(mx, 16));
        if (ti == 0 && tj * TS + y < k) cm0[tj * TS + y] = fmax(mx, cnorm[tj * TS + y]);
    }
    __syncthreads();
    // the largest of them: lets the per-pivot fast path test |a_jj| against ONE register value (a pivot that clears the largest
    // threshold clears its own); only candidates below it look their own threshold up (slow path)
    const int lane = tid & 63;
    return wave_max_all(fmax(lane < k ? cm0[lane] : 0.0, lane + 64 < k ? cm0[lane + 64] : 0.0));
}

// fronts of order 65 .. 128 with <= 16 pivots: static-order blocked elimination accepted a posteriori (defined behind the DPP helpers below)
__device__ __forceinline__ bool front_fast16(double* F, const int m, const int k, double* scratch, double* dinv_s, const double* cnorm, const double small, const double gmax, int& nneg);

// front kernel on the register-tiled core: LDS assembly (A scatter + children extend-add), tiles -> VGPRs, LDL^T,
// write-back of the pivot-ordered panel, the contribution block (straight from registers), pivot data and L11^{-1}.
// FAST (256 threads only): the fronts of the bucket with <= 16 pivots are assembled and eliminated on the static-order path (front_fast16) and
// marked done (hasis[s] = 2) when it accepts them; the ordinary instantiation is launched behind it with flags & 2 and leaves those alone.  Two
// kernels, not one with a branch: inlined into the strict kernel the fast path's 100 registers pushed <256, 6> from 134 to 459 spilled VGPRs.
template <int NT, int TS, bool FAST = false>
__global__ __launch_bounds__(NT, (NT == 64 && TS == 4) ? 4 : ((NT == 64 && TS == 2) ? 6 : ((NT == 256 && TS == 6) ? 3 : 1))) void k_front_reg(DevView V, int list_off, int flags)
{
    const int top_mode = flags & 1;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int G = (NT == 64) ? 8 : 16;
    constexpr int MAXM = G * TS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = NT / 64;
    const FrontMeta M = V.fmeta[list_off + blockIdx.x];
    const int s = M.s, c0 = M.c0, k = M.k, r0 = M.r0, m = M.m; (void)s; (void)c0; (void)r0; (void)k;
    if (FAST) { if (k > 16) { if (tid == 0) V.hasis[s] = 0; return; } }
    else if ((flags & 2) && V.hasis[s] == 2) return;
    const int ld = m | 1, ldi = k | 1;
    DBGSTAMP(4);
#ifdef MI355X_PIVSTAT
    long long ph0 = clock64(), ph1 = 0, ph2 = 0, ph3 = 0, ph4 = 0;
#endif
    // F: the front's LOWER triangle, packed by columns (column c starts at c m - c (c - 1) / 2) -- half the LDS of a square, so
    // twice as many fronts per CU on levels made of thousands of them (the pivot loop is a latency chain: co-resident fronts are
    // what fills the SIMDs); later overlaid by Lbuf (k columns of stride ld) + the k x k inverse
    const int npk = m * (m + 1) / 2;
    const int fdoubles = max(npk, k * ld + k * ldi);
    double* F      = reinterpret_cast<double*>(smem_raw);
    double* colbuf = F + fdoubles;               // 4 * MAXM
    double* dinv_s = colbuf + 4 * MAXM;          // k
    double* doff_s = dinv_s + k;                 // k
    double* cm0    = doff_s + k;                 // k   max |entry| of each fully-summed column of the assembled front
    int*    ord    = reinterpret_cast<int*>(cm0 + k);      // k
    int*    pt_s   = ord + k;                    // k
    auto pk = [m](int i, int c) { return c * m - ((c * (c - 1)) >> 1) + (i - c); };      // i >= c

    // ---- (a)-(c) assembly in LDS (lower storage) ----
    const bool from_arena = top_mode && V.arena && V.arena_off[s] >= 0;   // replicated front at a subtree join
    const bool skip_owned = top_mode && V.arena;
    if (from_arena) {
        const double* Ar = V.arena + V.arena_off[s];
        for (int idx = tid; idx < m * m; idx += NT) { int i = idx % m, c = idx / m; if (i >= c) F[pk(i, c)] = Ar[idx]; }
    } else {
        for (int idx = tid; idx < npk; idx += NT) F[idx] = 0.0;
    }
    __syncthreads();
    {
        const int q0 = M.aq0, q1 = M.aq1;
        for (int q = q0 + tid; q < q1; q += NT) { const int pos = V.apos[q]; const int i = pos % m, c = pos / m; F[pk(i, c)] += V.aval[q]; }
    }
    __syncthreads();
    // children: extend-add of each contribution block (lower triangle of an mc x mc square).  A dependent global access costs
    // 1.5-2 us on this part, so a child is NOT walked column by column: its lower triangle is enumerated flat -- column b paired
    // with column mc-1-b, mc+1 entries per pair -- and fetched in batches of 16 independent loads per thread, issued together
    // with the load of its relative indices; the scatter into F then runs out of registers.
    {
        int* relS = reinterpret_cast<int*>(colbuf);              // the child's relative indices (colbuf is free until the LDL^T)
        for (int cp = M.ch0; cp < M.ch1; ++cp) {
            const ChildMeta* Cp = V.cmeta + cp;
            if (skip_owned && Cp->owner >= 0) continue;
            const int mc = Cp->mc, ldt = Cp->ldt;
            const double* C = V.cb + Cp->cb_off;
            const int* relg = V.rel + Cp->relbase;
            const unsigned d = (unsigned)(mc + 1), inv = 0xFFFFFFFFu / d + 1u;      // floor(e / d) = umulhi(e, inv) for e < 2^16
            const int npr = (mc + 1) >> 1, total = npr * (mc + 1);
            int r0v = (tid < mc) ? relg[tid] : 0, r1v = 0;
            if (NT == 64 && tid + 64 < mc) r1v = relg[tid + 64];
            auto decode = [&](int e, int& a, int& b) -> bool {      // flat index -> (row a >= column b) of the child's lower triangle
                const int pr = (int)__umulhi((unsigned)e, inv), q = e - pr * (mc + 1);
                const bool second = q >= mc - pr;
                b = second ? mc - 1 - pr : pr;
                a = second ? b + (q - (mc - pr)) : pr + q;
                return e < total && !(second && 2 * pr == mc - 1);        // (mc odd: the middle column pairs with itself)
            };
            constexpr int U = (TS <= 6) ? 8 : 16;                     // loads in flight per thread (the small-tile instantiations are register-capped)
            for (int base = 0; base < total; base += NT * U) {
                double cvv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { int a, b; cvv[u] = decode(base + tid + NT * u, a, b) ? C[a + (size_t)b * ldt] : 0.0; }
                if (base == 0) {
                    if (tid < mc) relS[tid] = r0v;
                    if (NT == 64 && tid + 64 < mc) relS[tid + 64] = r1v;
                    __syncthreads();
                }
#pragma unroll
                for (int u = 0; u < U; ++u) { int a, b; if (decode(base + tid + NT * u, a, b)) F[pk(relS[a], relS[b])] += cvv[u]; }
            }
            __syncthreads();
        }
    }
#ifdef MI355X_PIVSTAT
    ph1 = clock64();
#endif
    // ---- fast path: <= 16 pivots eliminated in natural order by ONE wavefront (DPP), the rest of the front by MFMA; accepted a posteriori ----
    if constexpr (FAST) {
        int fneg = 0;
        const double gmax = 1.0 / fmax(fmax(V.pivtol, V.pivtol2), V.fastu);
        if (!front_fast16(F, m, k, colbuf, dinv_s, V.cnorm + c0, V.small, gmax, fneg)) { if (tid == 0) V.hasis[s] = 0; return; }
        {
            // F(i, c), c < k: W = L d (i > c);  F(i, c), i >= c >= k: the Schur complement;  colbuf: L11^{-1} (X(i, p) at [i + 17 p])
            double* Lg = V.L + M.panel_off;
            for (int c = wave; c < k; c += NW) {
                const double di = dinv_s[c];
                for (int i = lane; i < m; i += 64) Lg[i + (size_t)c * m] = (i > c) ? F[pk(i, c)] * di : 0.0;
            }
            for (int jj = tid; jj < k; jj += NT) { V.dinv[c0 + jj] = dinv_s[jj]; V.doff[c0 + jj] = 0.0; V.ptype[c0 + jj] = 1; V.lperm[c0 + jj] = jj; }
            const int mu = m - k;
            double* Cg = V.cb + M.cb_off;
            for (int c = k + wave; c < m; c += NW)
                for (int i = c + lane; i < m; i += 64) Cg[(i - k) + (size_t)(c - k) * mu] = F[pk(i, c)];
            if (tid == 0) V.fstat[s] = make_int4(fneg, 0, 0, 0);
            double* Mg = V.minv + M.minv_off;
            for (int idx = tid; idx < k * k; idx += NT) { const int i = idx % k, c = idx / k; Mg[idx] = (i > c) ? colbuf[i + 17 * c] : (i == c ? 1.0 : 0.0); }
            if (tid == 0) V.hasis[s] = 2;
            return;
        }
    }
    // ---- tiles -> registers (full symmetric) ----
    const int ti = tid % G, tj = tid / G, row0 = ti * TS, col0 = tj * TS;
    double t[TS][TS];
    {
        int rbase[TS];                                 // packed start of column (row0 + x), minus its index: F(c, i) for c > i sits at rbase[x] + c
#pragma unroll
        for (int x = 0; x < TS; ++x) { const int i = row0 + x; rbase[x] = i * m - ((i * (i - 1)) >> 1) - i; }
#pragma unroll
        for (int y = 0; y < TS; ++y) {
            const int c = col0 + y;
            const int cbase = c * m - ((c * (c - 1)) >> 1) - c;
#pragma unroll
            for (int x = 0; x < TS; ++x) {
                const int i = row0 + x;
                const int idx = (i >= c) ? cbase + i : rbase[x] + c;
                t[x][y] = (i < m && c < m) ? F[idx] : 0.0;
            }
        }
    }
    __syncthreads();                                   // F is dead from here on: its storage becomes Lbuf
    // ---- (d) LDL^T ----
    int nneg = 0, nzero = 0, ntwo = 0, nsmall = 0, chg = 0;
    DBGSTAMP(5);
#ifdef MI355X_PIVSTAT
    ph2 = clock64();
#endif
    const double cmx = front_colmax<NT, TS>(t, cm0, k, V.cnorm + c0);
    ldlt_reg<NT, TS, false>(t, m, k, F, ld, colbuf, dinv_s, doff_s, pt_s, ord, V.pivtol, V.pivtol2, V.small, cm0, cmx, V.zpiv + c0, nneg, nzero, ntwo, nsmall, chg);
    if (chg && tid == 0) V.qstat[0] = 1;
    __syncthreads();
    DBGSTAMP(6);
#ifdef MI355X_PIVSTAT
    ph3 = clock64();
#endif
    if (V.dbg && blockIdx.x == 0 && tid == 0) V.dbg[14] = (unsigned long long)k * 1000 + m;
    // ---- (e) write back: pivot-ordered panel, pivot data, contribution block from the registers ----
    double* Lg = V.L + M.panel_off;
    for (int c = wave; c < k; c += NW)
        for (int i = lane; i < m; i += 64) {
            const int src = (i < k) ? ord[i] : i;
            Lg[i + (size_t)c * m] = (i > c) ? F[src + c * ld] : 0.0;
        }
    for (int jj = tid; jj < k; jj += NT) { V.dinv[c0 + jj] = dinv_s[jj]; V.doff[c0 + jj] = doff_s[jj]; V.ptype[c0 + jj] = pt_s[jj]; V.lperm[c0 + jj] = ord[jj]; }
    const int mu = m - k;
    double* Cg = V.cb + M.cb_off;
#pragma unroll
    for (int y = 0; y < TS; ++y)
#pragma unroll
        for (int x = 0; x < TS; ++x) {
            const int i = row0 + x, c = col0 + y;
            if (c >= k && i >= c && i < m) Cg[(i - k) + (size_t)(c - k) * mu] = t[x][y];
        }
    if (tid == 0) V.fstat[s] = make_int4(nneg, nzero, ntwo, nsmall);
    // ---- (f) L11^{-1}: pivot-ordered unit-lower block built behind Lbuf, inverted in place ----
    double* Li = F + k * ld;
    for (int idx = tid; idx < k * k; idx += NT) { const int i = idx % k, c = idx / k; Li[i + c * ldi] = (i > c) ? F[ord[i] + c * ld] : 0.0; }
    __syncthreads();
    invert_unit_lower<NT>(Li, ldi, k);
    double* Mg = V.minv + M.minv_off;
    for (int idx = tid; idx < k * k; idx += NT) { const int i = idx % k, c = idx / k; Mg[idx] = (i > c) ? Li[i + c * ldi] : (i == c ? 1.0 : 0.0); }
#ifdef MI355X_PIVSTAT
    ph4 = clock64();
    if (NT == 256 && tid == 0) { const int o = (TS == 6) ? 16 : 24; atomicAdd(&g_fstat[o + 0], (unsigned long long)(ph1 - ph0)); atomicAdd(&g_fstat[o + 1], (unsigned long long)(ph2 - ph1)); atomicAdd(&g_fstat[o + 2], (unsigned long long)(ph3 - ph2)); atomicAdd(&g_fstat[o + 3], (unsigned long long)(ph4 - ph3)); atomicAdd(&g_fstat[o + 4], 1ull); atomicAdd(&g_fstat[o + 5], (unsigned long long)k); atomicAdd(&g_fstat[o + 6], (unsigned long long)(M.ch1 - M.ch0)); }
#endif
}

// ================================================================================================
// Blocked LDL^T of a pivot block with A POSTERIORI acceptance (the fast path of the big fronts' pivot blocks; SSIDS calls the
// idea a-posteriori threshold pivoting, IpSpralSolverInterface.cpp:199-204 "pivot method block").  The strict loop above decides
// one pivot at a time behind a workgroup barrier (~1 400 cycles per pivot, 255 CUs idle on the separator chains); here
//   * the k x k block sits in LDS; 16 columns at a time, ONE wavefront eliminates the 16 x 16 diagonal block in registers in natural
//     order with 1x1 pivots -- the pivot row travels by DPP inside the wavefront, no LDS, no barrier -- and inverts its unit-lower
//     factor the same way;
//   * the rows below and the remaining columns follow by v_mfma_f64_16x16x4_f64 (ldlt_blocked_static below);
//   * nothing is decided per pivot.  The block is ACCEPTED afterwards iff every pivot is clear of the zero threshold and every
//     multiplier of the block is <= gmax = 1 / max(u, u2, 1e-4) -- the threshold test at the LARGEST u Ipopt raises its solvers to (pivtolmax = 1e-4), far
//     tighter than the 1e-8 Ipopt asks for, so an accepted block satisfies the strict rule's tests at u AND at u2 (no u-sensitive
//     decision).  Otherwise nothing has been written and the caller runs the strict loop on the untouched block.
// Returns (workgroup-uniform) true when accepted: Lb = unit-lower L (strictly lower part, natural order), dinv_s = 1 / d.
// ================================================================================================
// ---- one 16 x 16 diagonal block in ONE wavefront, no LDS, no barrier: lane (i = lane & 15) holds row i, the four 16-lane DPP rows hold
// identical copies, a[c] = column c.  The pivot row travels inside the DPP row: v_fmac_f64_dpp ... row_newbcast:c (DP-ALU DPP supports
// exactly this control on gfx950) does  a[c] += a_j[lane c] * (-l)  in one instruction -- 15 - j instructions per pivot instead of a
// workgroup barrier, a column publication and 16 register-tile FMAs per thread. ----
template <int N> __device__ __forceinline__ double bcast16(const double x) { return __builtin_amdgcn_update_dpp(x, x, 0x150 + N, 0xF, 0xF, false); }      // row_newbcast:N
#define MI_FD(c) "v_fmac_f64_dpp %" #c ", %16, -%17 row_newbcast:" #c " row_mask:0xf bank_mask:0xf\n\t"
#define MI_R15 ""
#define MI_R14 MI_FD(15)
#define MI_R13 MI_FD(14) MI_R14
#define MI_R12 MI_FD(13) MI_R13
#define MI_R11 MI_FD(12) MI_R12
#define MI_R10 MI_FD(11) MI_R11
#define MI_R9 MI_FD(10) MI_R10
#define MI_R8 MI_FD(9) MI_R9
#define MI_R7 MI_FD(8) MI_R8
#define MI_R6 MI_FD(7) MI_R7
#define MI_R5 MI_FD(6) MI_R6
#define MI_R4 MI_FD(5) MI_R5
#define MI_R3 MI_FD(4) MI_R4
#define MI_R2 MI_FD(3) MI_R3
#define MI_R1 MI_FD(2) MI_R2
#define MI_R0 MI_FD(1) MI_R1
#define MI_AOPS "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
// a[c] -= t[lane c of the row] * l   for c = J+1 .. 15   (s_nop 1: a VALU-written VGPR needs two wait states before a DPP read, and the
// compiler does not pad inside an asm statement)
template <int J> __device__ __forceinline__ void rank1_dpp16(double (&a)[16], const double t, const double l)
{
#define MI_CASE(j) if constexpr (J == j) asm("s_nop 1\n\t" MI_R##j : MI_AOPS : "v"(t), "v"(l));
    MI_CASE(0) MI_CASE(1) MI_CASE(2) MI_CASE(3) MI_CASE(4) MI_CASE(5) MI_CASE(6) MI_CASE(7) MI_CASE(8) MI_CASE(9) MI_CASE(10) MI_CASE(11) MI_CASE(12) MI_CASE(13) MI_CASE(14)
#undef MI_CASE
}
// x += x[lane J of the row] * m   (one step of the row-oriented substitution for L^{-1})
template <int J> __device__ __forceinline__ void subst_dpp16(double& x, const double m)
{
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(m), "n"(J));
}
template <int J> __device__ __forceinline__ void diag16_step(double (&a)[16], double& rsave, const int l15)
{
    const double t = a[J];
    const double d = bcast16<J>(t);
    const double ri = fast_rcp(d);
    rsave = (l15 == J) ? ri : rsave;
    const double l = t * ri;
    rank1_dpp16<J>(a, t, l);
    a[J] = l;
}
template <int J> __device__ __forceinline__ void inv16_step(const double (&a)[16], double (&x)[4], const int l15)
{
    const double m = (l15 > J) ? -a[J] : 0.0;         // rows up to J are finished
    subst_dpp16<J>(x[0], m);
    if constexpr (J >= 4) subst_dpp16<J>(x[1], m);
    if constexpr (J >= 8) subst_dpp16<J>(x[2], m);
    if constexpr (J >= 12) subst_dpp16<J>(x[3], m);
}

// Blocked LDL^T of the k x k block in Lb (both triangles valid, ld), natural order, 16 columns at a time:
//   A  wavefront 0: the 16 x 16 diagonal block in registers (diag16_step), its inverse X = L^{-1} by the same DPP substitution, kept in
//      the operand layout of v_mfma_f64_16x16x4_f64 (lane (i, r) holds X(i, r), X(i, 4 + r), X(i, 8 + r), X(i, 12 + r));
//   B  the rows below, 16 per wavefront: W21 = A21 X^T (4 MFMAs), L21 = W21 D^{-1};
//   C  the trailing 16 x 16 tiles:  T(i, c) -= L21(i, :) W21(c, :)^T  (4 MFMAs each), spread over the wavefronts.
// Nothing is decided per pivot: the block is accepted A POSTERIORI (see above).  Isb receives the four X blocks (Isb[b*272 + i + p*17]):
// the blocked substitution of the panel rows (trsm_rows_impl) needs exactly these.
__device__ __forceinline__ bool ldlt_blocked_static(double* Lb, const int ld, const int k, double* Wb, double* dinv_s, double* Isb, int* shflag,
                                                    const double zmax, const double gmax, int& nneg, unsigned long long* dbg = nullptr, unsigned long long* ts = nullptr)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = (int)blockDim.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int nb = (k + 15) >> 4;
    if (tid == 0) *shflag = 0;
    __syncthreads();
    int neg = 0;
#define PSTAMP(i) do { if (dbg && b == 0 && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) dbg[40 + (i)] = clock64(); } while (0)
    for (int b = 0; b < nb; ++b) {
        const int c16 = 16 * b;
        if (wave == 0) {
            const int row = c16 + l15;
            double a[16];
            PSTAMP(0);
#pragma unroll
            for (int c = 0; c < 16; ++c) {                           // (identity beyond k; the load itself is unconditional: no branch per column)
                const double v = Lb[min(row, k - 1) + min(c16 + c, k - 1) * ld];
                a[c] = (row < k && c16 + c < k) ? v : ((l15 == c) ? 1.0 : 0.0);
            }
            double rsave = 1.0;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); PSTAMP(1);
            diag16_step<0>(a, rsave, l15); diag16_step<1>(a, rsave, l15); diag16_step<2>(a, rsave, l15); diag16_step<3>(a, rsave, l15);
            diag16_step<4>(a, rsave, l15); diag16_step<5>(a, rsave, l15); diag16_step<6>(a, rsave, l15); diag16_step<7>(a, rsave, l15);
            diag16_step<8>(a, rsave, l15); diag16_step<9>(a, rsave, l15); diag16_step<10>(a, rsave, l15); diag16_step<11>(a, rsave, l15);
            diag16_step<12>(a, rsave, l15); diag16_step<13>(a, rsave, l15); diag16_step<14>(a, rsave, l15); diag16_step<15>(a, rsave, l15);
            PSTAMP(2);
            // a posteriori: multipliers of the block, pivots against the zero threshold (|d| > zmax  <=>  |1/d| < 1/zmax; inf / NaN fail)
            double gm = 0.0;
#pragma unroll
            for (int c = 0; c < 15; ++c) gm = fmax(gm, (l15 > c) ? fabs(a[c]) : 0.0);
            const bool mine = row < k;
            bool bad = __ballot(gm > gmax) != 0ull;
            bad |= __ballot(mine && !(fabs(rsave) * zmax < 1.0)) != 0ull;
            neg += __popcll(__ballot(mine && l4 == 0 && rsave < 0.0));
            PSTAMP(3);
            double x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) x[q] = (l15 == 4 * q + l4) ? 1.0 : 0.0;
            inv16_step<0>(a, x, l15); inv16_step<1>(a, x, l15); inv16_step<2>(a, x, l15); inv16_step<3>(a, x, l15); inv16_step<4>(a, x, l15);
            inv16_step<5>(a, x, l15); inv16_step<6>(a, x, l15); inv16_step<7>(a, x, l15); inv16_step<8>(a, x, l15); inv16_step<9>(a, x, l15);
            inv16_step<10>(a, x, l15); inv16_step<11>(a, x, l15); inv16_step<12>(a, x, l15); inv16_step<13>(a, x, l15); inv16_step<14>(a, x, l15);
            PSTAMP(4);
            if (dbg && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) dbg[24 + 2 * b] = clock64();
            if (bad) { if (lane == 0) *shflag = 1; }
            else {
                if (l4 == 0 && mine) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) if (c16 + c < k) Lb[row + (c16 + c) * ld] = (l15 > c) ? a[c] : 0.0;
                    dinv_s[row] = rsave;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) Isb[b * 272 + l15 + (4 * q + l4) * 17] = x[q];
            }
            PSTAMP(5);
        }
        __syncthreads();
        if (ts) { if (b == 0) ts[16] = clock64(); if (b == 1) ts[19] = clock64(); if (b == 2) ts[22] = clock64(); if (b == 3) ts[25] = clock64(); }
        if (*shflag) return false;
        if (dbg && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) dbg[25 + 2 * b] = clock64();
        if (b + 1 >= nb) break;
        // ---- B: rows below the diagonal block ----
        for (int t = wave; t < nb - b - 1; t += nw) {
            const int r = c16 + 16 * (t + 1) + l15;
            v4f64_ acc = (v4f64_){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double av = Isb[b * 272 + l15 + (4 * u + l4) * 17];                       // X(c = l15, p = 4u + l4)
                const double bl = Lb[min(r, k - 1) + min(c16 + 4 * u + l4, k - 1) * ld];       // (unconditional load, clamped: no branch per operand)
                const double bv = (r < k && c16 + 4 * u + l4 < k) ? bl : 0.0;                    // A21(r, p)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
            bool big = false;
#pragma unroll
            for (int g = 0; g < 4; ++g) {                                                        // acc[g] = W21(r, c = l4 + 4g)
                const int c = l4 + 4 * g;
                const double l = acc[g] * dinv_s[c16 + c];
                Wb[(r - c16) + c * 65] = acc[g];
                if (r < k && c16 + c < k) { Lb[r + (c16 + c) * ld] = l; big |= fabs(l) > gmax; }
            }
            if (__ballot(big) != 0ull && lane == 0) *shflag = 1;
        }
        __syncthreads();
        if (ts) { if (b == 0) ts[17] = clock64(); if (b == 1) ts[20] = clock64(); if (b == 2) ts[23] = clock64(); }
        if (wave == 0) PSTAMP(6);
        if (*shflag) return false;
        // ---- C: rank-16 update of the trailing tiles: T(i, c) -= sum_p L(i, c16 + p) W(c, p) ----
        {
            int q = 0;
            for (int tc = b + 1; tc < nb; ++tc)
                for (int ti = tc; ti < nb; ++ti, ++q) {
                    if (q % nw != wave) continue;
                    v4f64_ acc = (v4f64_){0.0, 0.0, 0.0, 0.0};
                    const int ri = 16 * ti + l15;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const double al = Lb[min(ri, k - 1) + min(c16 + 4 * t + l4, k - 1) * ld];
                        const double av = (ri < k && c16 + 4 * t + l4 < k) ? al : 0.0;
                        const double bv = Wb[(16 * tc - c16 + l15) + (4 * t + l4) * 65];
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                    }
                    const int cc = 16 * tc + l15;
#pragma unroll
                    for (int g = 0; g < 4; ++g) { const int rr = 16 * ti + l4 + 4 * g; if (rr < k && cc < k) Lb[rr + cc * ld] -= acc[g]; }
                }
        }
        __syncthreads();
        if (ts) { if (b == 0) ts[18] = clock64(); if (b == 1) ts[21] = clock64(); if (b == 2) ts[24] = clock64(); }
        if (wave == 0) PSTAMP(7);
    }
#undef PSTAMP
    nneg += neg;
    return true;
}

// ================================================================================================
// Fronts of order 65 .. 128 with k <= 16 pivots (the bulk of the 256-thread front kernel's work on 2-D problems: 33 000 fronts per
// factorisation of synth_1e6, mean k = 15): the strict loop costs ~2 700 cycles per pivot there (three workgroups share a CU).  Same idea as
// the pivot blocks of the big fronts: natural order, 1x1 pivots, nothing decided per pivot, accepted A POSTERIORI --
//   A  wavefront 0: rows 0..15 of the front in registers, the k pivots eliminated by DPP (fast16_step; the non-pivot rows k..15 ride along and
//      come out holding their Schur complement entries), X = L11^{-1} by the same DPP substitution;
//   B  the rows below, 16 per wavefront and pass: W21 = A21 X^T by 4 MFMAs, multipliers W21 D^{-1} checked;
//   -- acceptance: every multiplier of the WHOLE front column (a front of this size sees all its rows) <= gmax, every pivot clear of the zero
//      threshold; on rejection NOTHING has been written into F and the caller runs the strict loop --
//   C  the trailing 16 x 16 tiles: S(i, c) -= sum_p L(i, p) W(c, p) (4 MFMAs each) on the packed lower triangle.
// F: the assembled front, lower triangle packed by columns.  On acceptance F(i, c) = W(i, c) = L(i, c) d_c for c < k < = i..., the Schur
// complement for i >= c >= k; dinv_s = 1 / d; scratch[i + 17 p] = X(i, p) (the inverse the triangular solves use).
// ================================================================================================
template <int J> __device__ __forceinline__ void fast16_step(double (&a)[16], double (&w)[16], double& rsave, const int l15)
{
    const double t = a[J];
    w[J] = t;
    const double d = bcast16<J>(t);
    const double ri = fast_rcp(d);
    rsave = (l15 == J) ? ri : rsave;
    const double l = t * ri;
    rank1_dpp16<J>(a, t, l);
    a[J] = l;
}
template <int J> __device__ __forceinline__ void fast16_inv(const double (&a)[16], double (&x)[4], const int l15, const int k)
{
    const double m = (l15 > J && l15 < k) ? -a[J] : 0.0;         // rows up to J are finished; rows from k on are not part of L11
    subst_dpp16<J>(x[0], m);
    if constexpr (J >= 4) subst_dpp16<J>(x[1], m);
    if constexpr (J >= 8) subst_dpp16<J>(x[2], m);
    if constexpr (J >= 12) subst_dpp16<J>(x[3], m);
}
__device__ __forceinline__ bool front_fast16(double* F, const int m, const int k, double* scratch, double* dinv_s, const double* cnorm, const double small, const double gmax, int& nneg)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    auto pk = [m](int i, int c) { return c * m - ((c * (c - 1)) >> 1) + (i - c); };      // i >= c
    double* Xs = scratch;                                        // 16 x 17
    int* shflag = reinterpret_cast<int*>(scratch + 280);         // [0] rejected, [1] negative pivots
    double* cmv = scratch + 288;                                 // 16 column maxima
    // zero threshold of the front: largest |entry| of the assembled pivot columns / of the input columns
    {
        const int c = tid >> 4, part = tid & 15;
        double mx = 0.0;
        if (c < k) for (int i = part; i < m; i += 16) mx = fmax(mx, fabs(F[pk(max(i, c), min(i, c))]));
        mx = fmax(mx, dpp_f64<0xB1>(mx)); mx = fmax(mx, dpp_f64<0x4E>(mx)); mx = fmax(mx, dpp_f64<0x141>(mx)); mx = fmax(mx, dpp_f64<0x140>(mx));
        if (part == 0) cmv[c] = (c < k) ? fmax(mx, cnorm[c]) : 0.0;
        if (tid == 0) { shflag[0] = 0; shflag[1] = 0; }
    }
    __syncthreads();
    double cmx = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) cmx = fmax(cmx, cmv[c]);
    const double zmax = fmax(small, ZERO_REL * cmx);
    // ---- A ----
    double a[16], w[16], rsave = 1.0;
    if (wave == 0) {
#pragma unroll
        for (int c = 0; c < 16; ++c) { a[c] = F[pk(max(l15, c), min(l15, c))]; w[c] = 0.0; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define MI_STEP(j) if (j < k) fast16_step<j>(a, w, rsave, l15);
        MI_STEP(0) MI_STEP(1) MI_STEP(2) MI_STEP(3) MI_STEP(4) MI_STEP(5) MI_STEP(6) MI_STEP(7)
        MI_STEP(8) MI_STEP(9) MI_STEP(10) MI_STEP(11) MI_STEP(12) MI_STEP(13) MI_STEP(14) MI_STEP(15)
#undef MI_STEP
        double gm = 0.0;
#pragma unroll
        for (int c = 0; c < 15; ++c) gm = fmax(gm, (c < k && l15 > c) ? fabs(a[c]) : 0.0);
        const bool mine = l15 < k;
        bool bad = __ballot(gm > gmax) != 0ull;
        bad |= __ballot(mine && !(fabs(rsave) * zmax < 1.0)) != 0ull;
        const int neg = __popcll(__ballot(mine && l4 == 0 && rsave < 0.0));
        double x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = (l15 == 4 * q + l4) ? 1.0 : 0.0;
#define MI_INV(j) if (j + 1 < k) fast16_inv<j>(a, x, l15, k);
        MI_INV(0) MI_INV(1) MI_INV(2) MI_INV(3) MI_INV(4) MI_INV(5) MI_INV(6) MI_INV(7) MI_INV(8) MI_INV(9) MI_INV(10) MI_INV(11) MI_INV(12) MI_INV(13) MI_INV(14)
#undef MI_INV
        if (bad) { if (lane == 0) shflag[0] = 1; }
        else {
#pragma unroll
            for (int q = 0; q < 4; ++q) Xs[l15 + (4 * q + l4) * 17] = x[q];      // X(i = l15, p = 4 q + l4)
            if (l4 == 0 && mine) dinv_s[l15] = rsave;
            if (lane == 0) shflag[1] = neg;
        }
    }
    __syncthreads();
    if (shflag[0]) return false;
    // ---- B: rows 16 .. m-1, W21 = A21 X^T (kept in registers until the front is accepted) ----
    const int ntl = (m + 15) >> 4;                               // 16-row tiles of the front (tile 0 = wavefront 0's rows)
    v4f64_ wacc[2];
    int wrow[2] = {-1, -1};
    {
        bool big = false;
        int slot = 0;
        for (int t = 1 + wave; t < ntl; t += 4, ++slot) {        // (m <= 128: at most 2 tiles per wavefront)
            const int r = 16 * t + l15;
            v4f64_ acc = (v4f64_){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = 4 * u + l4;
                const double av = (p < k) ? Xs[l15 + p * 17] : 0.0;                  // X(c = l15, p)   (zero beyond k: columns >= k are not pivots)
                const double bv = (r < m && p < k) ? F[pk(min(r, m - 1), p)] : 0.0;   // A21(r, p)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64((l15 < k) ? av : 0.0, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) { const int c = l4 + 4 * g; if (c < k && r < m) big |= fabs(acc[g] * dinv_s[c]) > gmax; }      // acc[g] = W21(r, c)
            if (slot == 0) { wacc[0] = acc; wrow[0] = r; } else { wacc[1] = acc; wrow[1] = r; }
        }
        if (__ballot(big) != 0ull && lane == 0) shflag[0] = 1;
    }
    __syncthreads();
    if (shflag[0]) return false;
    // ---- accepted: W into the first k columns of F, the Schur complement of rows k..15 ----
    nneg = shflag[1];
    if (wave == 0 && l4 == 0) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (c < k) { if (l15 > c) F[pk(l15, c)] = w[c]; }
            else if (l15 >= c) F[pk(l15, c)] = a[c];
        }
    }
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        const int r = wrow[sl];
        if (r >= 0 && r < m) {
#pragma unroll
            for (int g = 0; g < 4; ++g) { const int c = l4 + 4 * g; if (c < k) F[pk(r, c)] = wacc[sl][g]; }
        }
    }
    __syncthreads();
    // ---- C: S(i, c) -= sum_p L(i, p) W(c, p),  i >= 16, i >= c >= k ----
    {
        int q = 0;
        for (int tc = 0; tc < ntl; ++tc)
            for (int ti = max(tc, 1); ti < ntl; ++ti, ++q) {
                if ((q & 3) != wave) continue;
                v4f64_ acc = (v4f64_){0.0, 0.0, 0.0, 0.0};
                const int ri = 16 * ti + l15, rc = 16 * tc + l15;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int p = 4 * t + l4;
                    const double av = (ri < m && p < k) ? F[pk(min(ri, m - 1), p)] * dinv_s[min(p, 15)] : 0.0;       // L(ri, p)
                    const double bv = (rc < m && rc >= k && p < k) ? F[pk(min(max(rc, p), m - 1), p)] : 0.0;        // W(rc, p)   (rc >= k > p)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
                const int cc = 16 * tc + l15;
#pragma unroll
                for (int g = 0; g < 4; ++g) { const int rr = 16 * ti + l4 + 4 * g; if (rr < m && cc >= k && rr >= cc) F[pk(rr, cc)] -= acc[g]; }
            }
    }
    __syncthreads();
    return true;
}

// ================================================================================================
// Fronts of order <= 16 (the LukVl regime: 10^5 of them per tree level): FOUR fronts per wavefront, one per 16-lane DPP row, on the static-order
// path -- lane i of a row holds row i of its front in 16 registers, the pivots are eliminated by DPP (masked per front by its own pivot
// count), L11^{-1} by the same DPP substitution, everything accepted A POSTERIORI (every multiplier of the front <= gmax, every pivot clear
// of the zero threshold); a front that fails is left to the strict kernel launched behind (hasis[s] != 2).  The register-tiled kernel gives
// such a front a whole wavefront (8 x 8 threads with 2 x 2 tiles) and ~20 us of mostly latency; here four fronts share the latency and
// the instruction stream.
// ================================================================================================
template <int J> __device__ __forceinline__ void fast16_step_masked(double (&a)[16], double& rsave, const int l15, const bool on)
{
    const double t = a[J];
    const double d = bcast16<J>(t);
    const double ri = on ? fast_rcp(d) : 0.0;
    rsave = (on && l15 == J) ? ri : rsave;
    const double l = on ? t * ri : 0.0;
    rank1_dpp16<J>(a, t, l);
    a[J] = on ? l : t;
}
__global__ __launch_bounds__(64) void k_front_dpp16(DevView V, int list_off, int nfronts, int top_mode)
{
    __shared__ double Fs[4][16 * 17];
    __shared__ int relS[4][16];
    const int lane = threadIdx.x, g = lane >> 4, li = lane & 15;
    const int f = 4 * (int)blockIdx.x + g;
    const FrontMeta* Mp = V.fmeta + list_off + min(f, nfronts - 1);
    const int s = Mp->s, c0 = Mp->c0;
    const int m0 = Mp->m;
    const bool act = f < nfronts && m0 <= 16;                    // (larger fronts of the bucket: the strict kernel's)
    const int k = act ? Mp->k : 0, m = act ? m0 : 1;
    double* F = Fs[g];
    // ---- assembly: full symmetric 16 x 17 square per front ----
    const bool from_arena = act && top_mode && V.arena && V.arena_off[s] >= 0;
    const bool skip_owned = top_mode && V.arena;
    for (int idx = li; idx < 16 * 17; idx += 16) F[idx] = 0.0;
    if (from_arena) {
        const double* Ar = V.arena + V.arena_off[s];
        for (int idx = li; idx < m * m; idx += 16) { const int i = idx % m, c = idx / m; if (i >= c) { const double v = Ar[idx]; F[i + 17 * c] = v; F[c + 17 * i] = v; } }
    }
    if (act) {
        for (int q = Mp->aq0 + li; q < Mp->aq1; q += 16) {
            const int pos = V.apos[q]; const int i = pos % m, c = pos / m; const double v = V.aval[q];
            F[i + 17 * c] += v; if (i != c) F[c + 17 * i] += v;
        }
        for (int cp = Mp->ch0; cp < Mp->ch1; ++cp) {
            const ChildMeta* Cp = V.cmeta + cp;
            if (skip_owned && Cp->owner >= 0) continue;
            const int mc = Cp->mc, ldt = Cp->ldt;                  // (mc <= m <= 16: lane li owns row li of the child's lower triangle)
            const double* C = V.cb + Cp->cb_off;
            const bool mine = li < mc;
            if (mine) relS[g][li] = V.rel[Cp->relbase + li];
            double cv[16];
#pragma unroll
            for (int b = 0; b < 16; ++b) cv[b] = (mine && b <= li) ? C[li + (size_t)b * ldt] : 0.0;
            const int ra = relS[g][li];
#pragma unroll
            for (int b = 0; b < 16; ++b)
                if (mine && b <= li) { const int rb = relS[g][b]; F[ra + 17 * rb] += cv[b]; if (ra != rb) F[rb + 17 * ra] += cv[b]; }
        }
    }
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = F[li + 17 * c];
    // ---- zero threshold of the front: largest |entry| of its assembled pivot columns / of the input columns ----
    double cmine = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        double mx = fabs(a[c]);
        mx = fmax(mx, dpp_f64<0xB1>(mx)); mx = fmax(mx, dpp_f64<0x4E>(mx)); mx = fmax(mx, dpp_f64<0x141>(mx)); mx = fmax(mx, dpp_f64<0x140>(mx));
        cmine = (li == c) ? mx : cmine;
    }
    double cmx = (li < k) ? fmax(cmine, V.cnorm[c0 + li]) : 0.0;
    cmx = fmax(cmx, dpp_f64<0xB1>(cmx)); cmx = fmax(cmx, dpp_f64<0x4E>(cmx)); cmx = fmax(cmx, dpp_f64<0x141>(cmx)); cmx = fmax(cmx, dpp_f64<0x140>(cmx));
    const double zmax = fmax(V.small, ZERO_REL * cmx);
    const double gmax = 1.0 / fmax(fmax(V.pivtol, V.pivtol2), V.fastu);
    // ---- elimination (every front of the wavefront walks the same steps, masked by its own pivot count) ----
    double rsave = 1.0;
    const int kw = max(max(__builtin_amdgcn_readlane(k, 0), __builtin_amdgcn_readlane(k, 16)), max(__builtin_amdgcn_readlane(k, 32), __builtin_amdgcn_readlane(k, 48)));
#define MI_STEP(j) if (j < kw) fast16_step_masked<j>(a, rsave, li, j < k);
    MI_STEP(0) MI_STEP(1) MI_STEP(2) MI_STEP(3) MI_STEP(4) MI_STEP(5) MI_STEP(6) MI_STEP(7)
    MI_STEP(8) MI_STEP(9) MI_STEP(10) MI_STEP(11) MI_STEP(12) MI_STEP(13) MI_STEP(14) MI_STEP(15)
#undef MI_STEP
    double gm = 0.0;
#pragma unroll
    for (int c = 0; c < 15; ++c) gm = fmax(gm, (c < k && li > c) ? fabs(a[c]) : 0.0);
    const bool pv = li < k;
    const unsigned long long badm = __ballot(gm > gmax) | __ballot(pv && !(fabs(rsave) * zmax < 1.0));
    const unsigned long long negm = __ballot(pv && rsave < 0.0);
    const bool ok = act && ((badm >> (16 * g)) & 0xffffull) == 0ull;
    if (act && li == 0) V.hasis[s] = ok ? 2 : 0;
    // ---- L11^{-1}: row li of X in 16 registers, DPP substitution ----
    double x[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) x[p] = (p == li) ? 1.0 : 0.0;
#define MI_INV(j) if (j + 1 < kw) { const double mj = (li > j && li < k) ? -a[j] : 0.0; _Pragma("unroll") for (int p = 0; p <= j; ++p) subst_dpp16<j>(x[p], mj); }
    MI_INV(0) MI_INV(1) MI_INV(2) MI_INV(3) MI_INV(4) MI_INV(5) MI_INV(6) MI_INV(7) MI_INV(8) MI_INV(9) MI_INV(10) MI_INV(11) MI_INV(12) MI_INV(13) MI_INV(14)
#undef MI_INV
    if (!ok) return;
    // ---- results ----
    double* Lg = V.L + Mp->panel_off;
    double* Cg = V.cb + Mp->cb_off;
    double* Mg = V.minv + Mp->minv_off;
    const int mu = m - k;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        if (c < k) {
            if (li < m) Lg[li + (size_t)c * m] = (li > c) ? a[c] : 0.0;
            if (li < k) Mg[li + (size_t)c * k] = (li > c) ? x[c] : (li == c ? 1.0 : 0.0);
        } else if (li >= c && li < m) Cg[(li - k) + (size_t)(c - k) * mu] = a[c];
    }
    if (pv) { V.dinv[c0 + li] = rsave; V.doff[c0 + li] = 0.0; V.ptype[c0 + li] = 1; V.lperm[c0 + li] = li; }
    if (li == 0) V.fstat[s] = make_int4(__popcll((negm >> (16 * g)) & 0xffffull), 0, 0, 0);
}

// pivot block of a BIG front on the register-tiled core: 4x4 tiles on 16x16 threads for k <= 64, on 32x32 threads (two-word
// alive mask) for the 128-column panels of the wide_panels option (19 ms against 28 ms with 8x8 tiles on 256 threads, but
// still slower per column than two 64-column blocks: option off by default)
__device__ __forceinline__ void chain_signal(int* flag, const int epoch);
template <int NAP = 2> __device__ __forceinline__ void chain_wait(const int* flag, const int epoch, int* err);
// PRE: the caller has left the assembled, fully updated block in Lb (lower triangle) AND in the panel storage; late_*: a row block's L21 the
// caller kept back in LDS, stored once the flag is up (off the critical chain)
__device__ __forceinline__ void trsm_store_l(const DevView& V, const FrontMeta& M, const double* Lr, const int ibase, const int rlim);
template <int TS, int NT, bool PRE = false>
__device__ __forceinline__ void big_diag_body(const DevView& V, const FrontMeta& M, char* smem_raw, int* flag = nullptr, const int epoch = 0, unsigned long long* ts = nullptr,
                                              const double* late_Lr = nullptr, const FrontMeta& late_M = FrontMeta(), const int late_ibase = 0, const int late_rows = 0)
{
    constexpr int G = (NT == 1024) ? 32 : 16, MAXM = G * TS;
    const int tid = threadIdx.x;
    const int s = M.s, c0 = M.c0, k = M.k;
    const int ld = k | 1;
    double* Lb     = reinterpret_cast<double*>(smem_raw);   // k x k L columns (physical rows); later the pivot-ordered block / its inverse
    double* colbuf = Lb + (size_t)ld * k;
    double* dinv_s = colbuf + 4 * MAXM; double* doff_s = dinv_s + k; double* cm0 = doff_s + k;
    int* ord = reinterpret_cast<int*>(cm0 + k); int* pt_s = ord + k;
    double* P = V.L + M.panel_off;
    const size_t ldp = (size_t)M.ldp;
    int nneg = 0, nzero = 0, ntwo = 0, nsmall = 0, chg = 0;
    DBGSTAMP(0);
#ifdef MI355X_PIVSTAT
    const bool dprobe = gridDim.x == 1 && gridDim.y > 1 && tid == 0 && NT == 256;
    if (dprobe) g_dt[1] = wall_clock64();
#endif
    if (V.dbg && blockIdx.x == 0 && blockIdx.y < 4 && tid == 0) { V.dbg[32 + 8 * blockIdx.y + 5] = wall_clock64(); }
    bool fast = false;
    if constexpr (NT == 256) {
        if (V.fastpiv && k <= 64) {
            // ---- fast path: blocked LDL^T in natural order, accepted a posteriori (ldlt_blocked_static) ----
            double* Wp = reinterpret_cast<double*>(pt_s + k + (k & 1));         // 64 x 16 (ld 65): W = L D of the panel in flight
            double* Isb = Wp + 16 * 65;                                          // 4 x 272: inverses of the 16 x 16 diagonal blocks of L11
            int* shflag = reinterpret_cast<int*>(Isb + 4 * 272);
            const int i = tid & 63, cq = tid >> 6;
            DBGT(0);
            if constexpr (!PRE) {   // the block (lower part in the panel storage) -> LDS, mirrored; a chain link's own A entries are added in LDS
                double pv[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) { const int c = cq + 4 * e; pv[e] = (i < k && c < k && i >= c) ? P[i + (size_t)c * ldp] : 0.0; }
                int apos_[2]; double aval_[2]; int na = 0;
                if (M.selfasm) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) { const int q = M.aq0 + tid + e * NT; if (q < M.aq1) { apos_[e] = V.apos[q]; aval_[e] = V.aval[q]; na = e + 1; } }
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) { const int c = cq + 4 * e; if (i < k && c < k && i >= c) { Lb[i + c * ld] = pv[e]; Lb[c + i * ld] = pv[e]; } }
                __syncthreads();
                if (M.selfasm) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) if (e < na) { const int ii = apos_[e] % M.m, cc = apos_[e] / M.m; if (ii < k) { Lb[ii + cc * ld] += aval_[e]; if (ii != cc) Lb[cc + ii * ld] += aval_[e]; } }
                    for (int q = M.aq0 + tid + 2 * NT; q < M.aq1; q += NT) { const int pos = V.apos[q]; const int ii = pos % M.m, cc = pos / M.m; if (ii < k) { const double v = V.aval[q]; Lb[ii + cc * ld] += v; if (ii != cc) Lb[cc + ii * ld] += v; } }
                    __syncthreads();
                }
            }
            DBGT(1);
            if (ts) ts[8] = clock64();
            // zero-pivot scale: the largest column scale of the block (a pivot that clears it clears its own column's)
            double cmx;
            {
                const int c = tid >> 2, part = tid & 3;
                double mx = 0.0;
                if (c < k) for (int r = part; r < k; r += 4) mx = fmax(mx, fabs(PRE ? Lb[max(r, c) + min(r, c) * ld] : Lb[r + c * ld]));
                mx = fmax(mx, dpp_f64<0xB1>(mx)); mx = fmax(mx, dpp_f64<0x4E>(mx));
                if (c < k && part == 0) mx = fmax(mx, V.cnorm[c0 + c]); else if (c >= k) mx = 0.0;
                cmx = wave_max_all(mx);
                if ((tid & 63) == 0) colbuf[tid >> 6] = cmx;
                __syncthreads();
                cmx = fmax(fmax(colbuf[0], colbuf[1]), fmax(colbuf[2], colbuf[3]));
            }
            const double zmax = fmax(V.small, ZERO_REL * cmx);
            const double gmax = 1.0 / fmax(fmax(V.pivtol, V.pivtol2), V.fastu);
            DBGT(2);
            if (ts) ts[9] = clock64();
            fast = ldlt_blocked_static(Lb, ld, k, Wp, dinv_s, Isb, shflag, zmax, gmax, nneg, V.dbg, ts);
            DBGT(3);
            if (ts) ts[10] = clock64();
            if (fast) {
                DBGSTAMP(1);
#pragma unroll
                for (int e = 0; e < 16; ++e) { const int c = cq + 4 * e; if (i < k && c < k) { const double v = (i > c) ? Lb[i + c * ld] : 0.0; P[i + (size_t)c * ldp] = v; if (i <= c) Lb[i + c * ld] = 0.0; } }
                for (int j = tid; j < k; j += NT) { doff_s[j] = 0.0; pt_s[j] = 1; ord[j] = j; }
                {   // the diagonal-block inverses travel with L11: the panel solves need exactly these
                    double* Ig = V.isg + (size_t)M.bigidx * ISG_STRIDE;
                    const int nis = ((k + 15) >> 4) * 272;
                    for (int idx = tid; idx < nis; idx += NT) Ig[idx] = Isb[idx];
                }
                if (tid == 0) atomicAdd(&V.qstat[3], 1);
                __syncthreads();
            } else if (tid == 0) atomicAdd(&V.qstat[2], 1);
        }
    }
    if (!fast) {
    if (M.selfasm) {            // pure in-place chain link (no assembly launch): the A entries of the pivot rows are added here
        for (int q = M.aq0 + tid; q < M.aq1; q += NT) { const int pos = V.apos[q]; const int i = pos % M.m, c = pos / M.m; if (i < k) P[i + (size_t)c * ldp] += V.aval[q]; }
        __syncthreads();
    }
    const int ti = tid % G, tj = tid / G, row0 = ti * TS, col0 = tj * TS;
    double t[TS][TS];
#pragma unroll
    for (int x = 0; x < TS; ++x)
#pragma unroll
        for (int y = 0; y < TS; ++y) {
            const int i = row0 + x, c = col0 + y;
            t[x][y] = (i < k && c < k) ? ((i >= c) ? P[i + (size_t)c * ldp] : P[c + (size_t)i * ldp]) : 0.0;
        }
    nneg = 0;
    const double cmx = front_colmax<NT, TS>(t, cm0, k, V.cnorm + c0);
    ldlt_reg<NT, TS, (G * TS > 64)>(t, k, k, Lb, ld, colbuf, dinv_s, doff_s, pt_s, ord, V.pivtol, V.pivtol2, V.small, cm0, cmx, V.zpiv + c0, nneg, nzero, ntwo, nsmall, chg);
    if (chg && tid == 0) V.qstat[0] = 1;
    __syncthreads();
    DBGSTAMP(1);
#ifdef MI355X_PIVSTAT
    if (dprobe) g_dt[2] = wall_clock64();
#endif
    // pivot-ordered unit-lower block: the row permutation is done IN PLACE in LDS through registers (each thread owns
    // <= 16 entries: k*k <= 16*NT), the panel gets its copy on the way -- no second k x k buffer, no global round trip
    {
        double tmp[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) { const int idx = tid + e * NT; const int i = idx % k, c = idx / k; tmp[e] = (idx < k * k && i > c) ? Lb[ord[i] + c * ld] : 0.0; }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) { const int idx = tid + e * NT; const int i = idx % k, c = idx / k; if (idx < k * k) { Lb[i + c * ld] = tmp[e]; P[i + (size_t)c * ldp] = tmp[e]; } }
    }
    }
    for (int j = tid; j < k; j += NT) { V.dinv[c0 + j] = dinv_s[j]; V.doff[c0 + j] = doff_s[j]; V.ptype[c0 + j] = pt_s[j]; V.lperm[c0 + j] = ord[j]; }
    if (tid == 0) { V.fstat[s] = make_int4(nneg, nzero, ntwo, nsmall); V.hasis[s] = fast ? 1 : 0; }
#ifdef MI355X_PIVSTAT
    if (flag && dprobe) g_dt[5] = wall_clock64();
#endif
    if (ts) ts[11] = clock64();
    if (flag) chain_signal(flag, epoch);        // fused launch: the panel workgroups need L11, D and the pivot order -- not the inverse below
    if (ts) ts[12] = clock64();
    if (late_Lr) trsm_store_l(V, late_M, late_Lr, late_ibase, late_rows);
    if (V.dbg && blockIdx.x == 0 && blockIdx.y < 4 && tid == 0) { V.dbg[32 + 8 * blockIdx.y + 6] = wall_clock64(); }
    __syncthreads();
    DBGSTAMP(2);
#ifdef MI355X_PIVSTAT
    if (dprobe) g_dt[3] = wall_clock64();
#endif
    invert_unit_lower<NT>(Lb, ld, k);
    DBGSTAMP(3);
#ifdef MI355X_PIVSTAT
    if (dprobe) g_dt[4] = wall_clock64();
#endif
    double* Mg = V.minv + M.minv_off;
    for (int idx = tid; idx < k * k; idx += NT) { const int i = idx % k, c = idx / k; Mg[idx] = (i > c) ? Lb[i + c * ld] : (i == c ? 1.0 : 0.0); }
    if (V.dbg && blockIdx.x == 0 && tid == 0) V.dbg[15] = (unsigned long long)k;
}

// (host side: bytes of big_diag_body's LDS layout; maxm = 64 for the 256-thread, 128 for the 1024-thread instantiation)
static size_t diag_lds_bytes(int kk, int maxm)
{
    return (size_t)((kk | 1) * kk + 4 * maxm + 3 * kk) * sizeof(double) + (size_t)(2 * kk + 2) * sizeof(int) + (size_t)(16 * 65 + 4 * 272) * sizeof(double) + 64;
}
template <int TS, int NT = 256>
__global__ __launch_bounds__(NT) void k_big_diag_reg(DevView V, int list_off)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const FrontMeta M = V.fmeta[list_off + blockIdx.x];
    big_diag_body<TS, NT>(V, M, smem_raw);
}

// ------------------------------------------------------------------------------------------------
// inertia / statistics reduction (fixed order => deterministic)
// ------------------------------------------------------------------------------------------------
__global__ void k_reduce_stats(const int4* fstat, const int* owner, int nsn, int rank_filter, int* out)
{
    // integer sums: order-independent, so a grid of partial sums + atomicAdd stays deterministic
    __shared__ int sh[4][256];
    int a = 0, b = 0, c = 0, d = 0;
    for (int s = blockIdx.x * 256 + threadIdx.x; s < nsn; s += gridDim.x * 256) {
        if (rank_filter >= -1 && owner[s] != rank_filter) continue;
        const int4 v = fstat[s]; a += v.x; b += v.y; c += v.z; d += v.w;
    }
    sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = b; sh[2][threadIdx.x] = c; sh[3][threadIdx.x] = d;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (threadIdx.x < off) for (int q = 0; q < 4; ++q) sh[q][threadIdx.x] += sh[q][threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x < 4 && sh[threadIdx.x][0] != 0) atomicAdd(&out[threadIdx.x], sh[threadIdx.x][0]);
}
__global__ void k_zero_i32(int* p, int n) { if (threadIdx.x < n) p[threadIdx.x] = 0; }
// start of a factorisation in one launch: per-column flags cleared, quality flags cleared, the epoch of the flag-synchronised
// launches advanced
__global__ void k_factor_prologue(DevView V)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < V.n; i += gridDim.x * blockDim.x) { V.colfail[i] = 0; V.zpiv[i] = 0; }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < V.nsn; i += gridDim.x * blockDim.x) V.tcnt[i] = 0;
    if (blockIdx.x == 0 && threadIdx.x < 4) V.qstat[threadIdx.x] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) V.sepoch[2] += 1;
}
__global__ void k_fill_i32(int* p, int v, int n) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v; }

// ------------------------------------------------------------------------------------------------
// solves
// ------------------------------------------------------------------------------------------------
__global__ void k_load_rhs(DevView V, const double* b)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < V.n; i += gridDim.x * blockDim.x) V.xw[i] = V.scale[i] * b[V.perm[i]];
}
__global__ void k_store_sol(DevView V, double* b)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < V.n; i += gridDim.x * blockDim.x) b[V.perm[i]] = V.scale[i] * V.xw[i];
}

// iterative refinement in the scaled, permuted space:  xacc += xw;  xw <- bw - K xacc   (K = scaled matrix, symmetric row
// view => gather, no atomics).  first != 0: xacc = xw (no accumulation yet).  The residual is then solved for again.
__global__ void k_refine_residual(DevView V, int first)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < V.n; i += gridDim.x * blockDim.x)
        V.xacc[i] = first ? V.xw[i] : V.xacc[i] + V.xw[i];
}
__global__ void k_refine_spmv(DevView V)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < V.n; i += gridDim.x * blockDim.x) {
        double acc = 0.0;
        for (int p = V.rslot_ptr[i]; p < V.rslot_ptr[i + 1]; ++p) {
            const int q = V.rslot_idx[p];
            const int r = V.arow[q], c = V.acol[q];
            acc += V.aval[q] * V.xacc[r == i ? c : r];
        }
        V.xw[i] = V.bw[i] - acc;
    }
}
__global__ void k_save_rhs(DevView V) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < V.n; i += gridDim.x * blockDim.x) V.bw[i] = V.xw[i]; }
__global__ void k_refine_finish(DevView V) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < V.n; i += gridDim.x * blockDim.x) V.xw[i] += V.xacc[i]; }

// forward: y = L11^{-1} P b for the pivot rows (a k x k mat-vec with the stored inverse: no substitution chain),
// z = D^{-1} y, and the contribution  c = (children) - L21 y  for the ancestors is left in cvec (the parent
// gathers it: no atomics, deterministic).  One workgroup per front of order <= 128 (BIG fronts: k_fwd_grp).
template <int NT>
__global__ __launch_bounds__(NT) void k_fwd(DevView V, int list_off, int top_mode)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x;
    const FrontMeta M = V.fmeta[list_off + blockIdx.x];
    const int s = M.s, c0 = M.c0, k = M.k, r0 = M.r0, m = M.m; (void)s; (void)c0; (void)r0; (void)k;
    double* xp = reinterpret_cast<double*>(smem_raw);   // k   pivot rows (original local order)
    double* ys = xp + k;                                // k
    double* bp = ys + k;                                // k   pivot rows in pivot order
    double* xu = bp + k;                                // m-k update-row accumulators (LDS classes only)
    for (int i = tid; i < k; i += NT) xp[i] = V.xw[c0 + i];
    for (int i = k + tid; i < m; i += NT) xu[i - k] = 0.0;
    if (top_mode && V.top_rhs) {
        const double* tr = V.top_rhs + V.top_rhs_off[s];
        __syncthreads();
        for (int i = tid; i < m; i += NT) { if (i < k) xp[i] += tr[i]; else xu[i - k] += tr[i]; }
    }
    __syncthreads();
    for (int cp = M.ch0; cp < M.ch1; ++cp) {
        const ChildMeta Cm = V.cmeta[cp];
        const int ch = Cm.ch; (void)ch;
        if (top_mode && V.top_rhs && Cm.owner >= 0) continue;
        const int base = Cm.relbase, mc = Cm.mc;
        for (int t = tid; t < mc; t += NT) {
            const int tg = V.rel[base + t]; const double v = V.cvec[Cm.cvbase + t];
            if (tg < k) xp[tg] += v; else xu[tg - k] += v;
        }
        __syncthreads();
    }
    // y = Minv * (P b): thread j, independent (pipelined) loads down row j of the column-major inverse
    const double* Mg = V.minv + M.minv_off;
    for (int j = tid; j < k; j += NT) bp[j] = xp[V.lperm[c0 + j]];
    __syncthreads();
    for (int j = tid; j < k; j += NT) {
        double a0 = 0.0, a1 = 0.0;
        int p = 0;
        for (; p + 1 <= j; p += 2) { a0 += Mg[j + (size_t)p * k] * bp[p]; a1 += Mg[j + (size_t)(p + 1) * k] * bp[p + 1]; }
        if (p <= j) a0 += Mg[j + (size_t)p * k] * bp[p];
        ys[j] = a0 + a1;
    }
    __syncthreads();
    {
        const double* Lg = V.L + M.panel_off;
        for (int i = k + tid; i < m; i += NT) {
            double t0 = 0.0, t1 = 0.0;
            int j = 0;
            for (; j + 1 < k; j += 2) { t0 += Lg[i + (size_t)j * M.ldp] * ys[j]; t1 += Lg[i + (size_t)(j + 1) * M.ldp] * ys[j + 1]; }
            if (j < k) t0 += Lg[i + (size_t)j * M.ldp] * ys[j];
            V.cvec[M.cv + i] = xu[i - k] - (t0 + t1);
        }
    }
    for (int j = tid; j < k; j += NT) {
        const int pt = V.ptype[c0 + j];
        double z;
        if (pt == 1) z = ys[j] * V.dinv[c0 + j];
        else if (pt == 2) z = V.dinv[c0 + j] * ys[j] + V.doff[c0 + j] * ys[j + 1];
        else z = V.doff[c0 + j - 1] * ys[j - 1] + V.dinv[c0 + j] * ys[j];
        V.zb[c0 + j] = z;
    }
}

// backward: x_piv = P^T L11^{-T} ( z - L21^T x_upd ), written un-permuted
template <int NT>
__global__ __launch_bounds__(NT) void k_bwd(DevView V, int list_off)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = NT / 64;
    const FrontMeta M = V.fmeta[list_off + blockIdx.x];
    const int s = M.s, c0 = M.c0, k = M.k, r0 = M.r0, m = M.m; (void)s; (void)c0; (void)r0; (void)k;
    double* ws = reinterpret_cast<double*>(smem_raw);   // k
    double* xu = ws + k;                                // m-k gathered ancestor values (LDS classes)
    for (int i = k + tid; i < m; i += NT) xu[i - k] = V.xw[V.sn_rows[r0 + i]];
    for (int j = tid; j < k; j += NT) ws[j] = V.zb[c0 + j];
    __syncthreads();
    {
        const double* Lg = V.L + M.panel_off;
        for (int j = wave; j < k; j += NW) {
            double t = 0.0;
            for (int i = lane; i < m - k; i += 64) t += Lg[k + i + (size_t)j * M.ldp] * xu[i];
            t = wave_sum(t);
            if (lane == 0) ws[j] -= t;
        }
    }
    __syncthreads();
    // x_p = sum_{j >= p} Minv(j,p) w_j : thread p walks its own (contiguous) column of the inverse
    const double* Mg = V.minv + M.minv_off;
    for (int p = tid; p < k; p += NT) {
        double a0 = 0.0, a1 = 0.0;
        int j = p;
        for (; j + 1 < k; j += 2) { a0 += Mg[j + (size_t)p * k] * ws[j]; a1 += Mg[j + 1 + (size_t)p * k] * ws[j + 1]; }
        if (j < k) a0 += Mg[j + (size_t)p * k] * ws[j];
        V.xw[c0 + V.lperm[c0 + p]] = a0 + a1;
    }
}



// ------------------------------------------------------------------------------------------------
// Fronts of order <= 32 (the LukVl regime: 10^5..10^6 of them per sweep): TWO fronts per wavefront (one per half: a front of
// order <= 32 leaves half a wavefront idle) and three dependent memory phases instead of seven -- everything that only depends on
// the front record (pivot order, D, the row of L11^{-1} and the panel row / column a lane owns) is loaded up front into registers,
// then the children's contributions (forward) or the ancestors' solution entries (backward), then arithmetic.  The sweeps are
// bound by (dependent round trips) x (fronts / resident wavefronts), not by bytes.  KP = compile-time bound on the pivot count.
// ------------------------------------------------------------------------------------------------
template <int KP, int LPF = 32>      // LPF lanes per front: 32 (order <= 32, two fronts per wavefront) or 16 (levels whose fronts all have order <= 16: four)
__global__ __launch_bounds__(64) void k_fwd_pair(DevView V, int list_off, int nfronts)
{
    constexpr int FPW = 64 / LPF;
    __shared__ double xs[FPW][LPF], bp[FPW][LPF], ys[FPW][LPF + 1];
    const int lane = threadIdx.x, h = lane / LPF, li = lane % LPF;
    const int f = FPW * (int)blockIdx.x + h;
    const bool act = f < nfronts;
    const FrontMeta* Mp = V.fmeta + list_off + (act ? f : 0);
    const int c0 = Mp->c0, k = act ? Mp->k : 0, m = act ? Mp->m : 0, ch0 = Mp->ch0, ch1 = act ? Mp->ch1 : Mp->ch0, ldp = Mp->ldp;
    const long long cvo = Mp->cv;
    // ---- phase 1: depends on the front record only ----
    const bool piv = li < k, upd = li >= k && li < m;
    const double xwv = piv ? V.xw[c0 + li] : 0.0;
    const int lpv = piv ? V.lperm[c0 + li] : 0;
    int pt = 1; double dq = 0.0, oq = 0.0, oq1 = 0.0;
    if (piv) { pt = V.ptype[c0 + li]; dq = V.dinv[c0 + li]; oq = V.doff[c0 + li]; oq1 = li > 0 ? V.doff[c0 + li - 1] : 0.0; }
    const double* Mg = V.minv + Mp->minv_off;
    const double* Lg = V.L + Mp->panel_off;
    double mrow[KP], lrow[KP];
#pragma unroll
    for (int p = 0; p < KP; ++p) {
        mrow[p] = (piv && p <= li) ? Mg[li + (size_t)p * k] : 0.0;
        lrow[p] = (upd && p < k) ? Lg[li + (size_t)p * ldp] : 0.0;
    }
    int cmc[4], crel[4]; long long ccv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const bool has = ch0 + c < ch1;
        const ChildMeta* Cp = V.cmeta + ch0 + (has ? c : 0);
        cmc[c] = has ? Cp->mc : 0; crel[c] = Cp->relbase; ccv[c] = Cp->cvbase;
    }
    // ---- phase 2: the children's contribution vectors ----
    int tg[4]; double cv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { const bool on = li < cmc[c]; tg[c] = on ? V.rel[crel[c] + li] : -1; cv[c] = on ? V.cvec[ccv[c] + li] : 0.0; }
    // ---- phase 3: arithmetic ----
    xs[h][li] = xwv;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) { if (tg[c] >= 0) xs[h][tg[c]] += cv[c]; __syncthreads(); }
    for (int cp = ch0 + 4; cp < ch1; ++cp) {                 // (more than four children: rare)
        const int mc = V.cmeta[cp].mc;
        if (li < mc) xs[h][V.rel[V.cmeta[cp].relbase + li]] += V.cvec[V.cmeta[cp].cvbase + li];
        __syncthreads();
    }
    bp[h][li] = piv ? xs[h][lpv] : 0.0;
    __syncthreads();
    double y = 0.0;
#pragma unroll
    for (int p = 0; p < KP; ++p) y += mrow[p] * bp[h][p];
    ys[h][li] = piv ? y : 0.0;
    if (li == 0) ys[h][LPF] = 0.0;
    __syncthreads();
    if (upd) {
        double t = 0.0;
#pragma unroll
        for (int p = 0; p < KP; ++p) t += lrow[p] * ys[h][p];
        V.cvec[cvo + li] = xs[h][li] - t;
    }
    if (piv) {
        double z;
        if (pt == 1) z = y * dq;
        else if (pt == 2) z = dq * y + oq * ys[h][li + 1];
        else z = oq1 * ys[h][li - 1] + dq * y;
        V.zb[c0 + li] = z;
    }
}
template <int KP, int LPF = 32>
__global__ __launch_bounds__(64) void k_bwd_pair(DevView V, int list_off, int nfronts)
{
    constexpr int FPW = 64 / LPF;
    __shared__ double xus[FPW][LPF], w[FPW][LPF];
    const int lane = threadIdx.x, h = lane / LPF, li = lane % LPF;
    const int f = FPW * (int)blockIdx.x + h;
    const bool act = f < nfronts;
    const FrontMeta* Mp = V.fmeta + list_off + (act ? f : 0);
    const int c0 = Mp->c0, k = act ? Mp->k : 0, m = act ? Mp->m : 0, r0 = Mp->r0, ldp = Mp->ldp;
    const bool piv = li < k, upd = li >= k && li < m;
    // ---- phase 1 ----
    const double zbv = piv ? V.zb[c0 + li] : 0.0;
    const int lpv = piv ? V.lperm[c0 + li] : 0;
    const int ridx = upd ? V.sn_rows[r0 + li] : 0;
    const double* Mg = V.minv + Mp->minv_off;
    const double* Lg = V.L + Mp->panel_off;
    double mcol[KP], lcol[LPF];
#pragma unroll
    for (int q = 0; q < KP; ++q) mcol[q] = (piv && li + q < k) ? Mg[li + q + (size_t)li * k] : 0.0;      // Minv(li + q, li)
#pragma unroll
    for (int i = 0; i < LPF; ++i) lcol[i] = (piv && k + i < m) ? Lg[k + i + (size_t)li * ldp] : 0.0;        // L(k + i, li)
    // ---- phase 2: the ancestors' solution entries ----
    xus[h][li] = upd ? V.xw[ridx] : 0.0;
    __syncthreads();
    // ---- phase 3 ----
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < LPF; ++i) t += lcol[i] * xus[h][(k + i) & (LPF - 1)];
    w[h][li] = piv ? zbv - t : 0.0;
    __syncthreads();
    if (piv) {
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < KP; ++q) a += mcol[q] * w[h][(li + q) & (LPF - 1)];
        V.xw[c0 + lpv] = a;
    }
}

// ================================================================================================
// BIG fronts in the triangular solves: a CHAIN GROUP (<= 4 links of an in-place separator chain, <= 256 columns) is one
// unit, handled at its LAST link (FrontMeta::grem == 0; the other links return at once).  All links of a chain share one
// forward vector (cvec + cv): the update entries of a link ARE the entries of the next link, nothing is copied.
//   k_fwd_grp      one workgroup: children gathered into the chain vector; per link  y = L11^{-1} P b, z = D^{-1} y, and
//                  the entries of the group's later pivots updated (<= 192 rows)
//   k_fwd_grp_upd  256 rows per workgroup: entries beyond the group  -=  sum over links  L21 y   (<= 256 columns in one pass)
//   k_bwd_grp_dot  256 rows per workgroup: partial  L21^T x  of the rows beyond the group, for all the group's columns
//   k_bwd_grp      one workgroup: per link (last to first)  x = P^T L11^{-T} ( z - partials - L(group rows)^T x )
// ================================================================================================
__global__ __launch_bounds__(256) void k_fwd_grp(DevView V, int list_off, int top_mode)
{
    __shared__ double bp[128], ys[128];
    const FrontMeta M = V.fmeta[list_off + blockIdx.x];
    if (M.grem != 0) return;
    const int tid = threadIdx.x;
    const int nl = M.gpos + 1;
    for (int j = 0; j < nl; ++j) {
        const GroupLink G = V.gtab[M.gbase + j];
        double* cv = V.cvec + G.cv;
        if (!G.alias) { for (int i = tid; i < G.m; i += 256) cv[i] = 0.0; __syncthreads(); }
        if (top_mode && V.top_rhs && G.tr >= 0) {
            const double* tr = V.top_rhs + G.tr;
            for (int i = tid; i < G.m; i += 256) cv[i] += tr[i];
            __syncthreads();
        }
        for (int cp = G.ch0; cp < G.ch1; ++cp) {
            const ChildMeta Cm = V.cmeta[cp];
            if (Cm.aliased) continue;
            if (top_mode && V.top_rhs && Cm.owner >= 0) continue;
            const int base = Cm.relbase, mc = Cm.mc;
            for (int t = tid; t < mc; t += 256) cv[V.rel[base + t]] += V.cvec[Cm.cvbase + t];
            __syncthreads();
        }
    }
    int done = 0;
    for (int j = 0; j < nl; ++j) {
        const GroupLink G = V.gtab[M.gbase + j];
        double* cv = V.cvec + G.cv;
        const int k = G.k, c0 = G.c0;
        done += k;
        const int rem = M.gcols - done;
        for (int p = tid; p < k; p += 256) { const int lp = V.lperm[c0 + p]; bp[p] = V.xw[c0 + lp] + cv[lp]; }
        __syncthreads();
        const double* Mg = V.minv + G.minv_off;
        {   // y = Minv (P b): 4 threads per row (columns p = part, part + 4, ...), combined through LDS
            const int q = tid >> 2, part = tid & 3;
            double a = 0.0;
            for (int q2 = q; q2 < k; q2 += 64) {
                a = 0.0;
                int p = part;
                for (; p + 12 <= q2; p += 16) {
                    const double m0 = Mg[q2 + (size_t)p * k], m1 = Mg[q2 + (size_t)(p + 4) * k], m2 = Mg[q2 + (size_t)(p + 8) * k], m3 = Mg[q2 + (size_t)(p + 12) * k];
                    a += m0 * bp[p] + m1 * bp[p + 4] + m2 * bp[p + 8] + m3 * bp[p + 12];
                }
                for (; p <= q2; p += 4) a += Mg[q2 + (size_t)p * k] * bp[p];
                a += __shfl_xor(a, 1); a += __shfl_xor(a, 2);
                if (part == 0) ys[q2] = a;
            }
        }
        __syncthreads();
        for (int q = tid; q < k; q += 256) {
            const int pt = V.ptype[c0 + q];
            double z;
            if (pt == 1) z = ys[q] * V.dinv[c0 + q];
            else if (pt == 2) z = V.dinv[c0 + q] * ys[q] + V.doff[c0 + q] * ys[q + 1];
            else z = V.doff[c0 + q - 1] * ys[q - 1] + V.dinv[c0 + q] * ys[q];
            V.zb[c0 + q] = z;
            V.ybuf[c0 + q] = ys[q];
        }
        const double* Lg = V.L + G.panel_off;
        for (int i = k + tid; i < k + rem; i += 256) {
            double t0 = 0.0, t1 = 0.0;
            int p = 0;
            for (; p + 7 < k; p += 8) {                      // 8 independent loads in flight per thread
                double l[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) l[u] = Lg[i + (size_t)(p + u) * G.ldp];
#pragma unroll
                for (int u = 0; u < 8; u += 2) { t0 += l[u] * ys[p + u]; t1 += l[u + 1] * ys[p + u + 1]; }
            }
            for (; p < k; ++p) t0 += Lg[i + (size_t)p * G.ldp] * ys[p];
            cv[i] -= t0 + t1;
        }
        __syncthreads();
    }
}

// Fused forward step of a one-link solve unit that has nothing to gather (FrontMeta::solo): block 0 does the pivot part,
// blocks 1.. each take 64 update rows and recompute y = L11^{-1} P b themselves (64 x 64 mat-vec from L2) instead of
// waiting for another launch.  b is read from xw, z goes to zb: nothing a sibling block reads is overwritten.
__global__ __launch_bounds__(256) void k_fwd_solo(DevView V, int list_off)
{
    __shared__ double bp[128], ys[128];
    __shared__ double red[4][64];
    const FrontMeta M = V.fmeta[list_off + blockIdx.y];
    const int k = M.k, m = M.m, c0 = M.c0;
    const int rb = (int)blockIdx.x - 1;                       // -1: pivot part
    if (rb >= 0 && k + rb * 64 >= m) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* cv = V.cvec + M.cv;
    for (int p = tid; p < k; p += 256) { const int lp = V.lperm[c0 + p]; bp[p] = V.xw[c0 + lp] + (M.solo == 1 ? cv[lp] : 0.0); }
    __syncthreads();
    const double* Mg = V.minv + M.minv_off;
    {
        const int part = tid & 3;
        for (int q = tid >> 2; q < k; q += 64) {
            double a = 0.0;
            int p = part;
            for (; p + 12 <= q; p += 16) {
                const double m0 = Mg[q + (size_t)p * k], m1 = Mg[q + (size_t)(p + 4) * k], m2 = Mg[q + (size_t)(p + 8) * k], m3 = Mg[q + (size_t)(p + 12) * k];
                a += m0 * bp[p] + m1 * bp[p + 4] + m2 * bp[p + 8] + m3 * bp[p + 12];
            }
            for (; p <= q; p += 4) a += Mg[q + (size_t)p * k] * bp[p];
            a += __shfl_xor(a, 1); a += __shfl_xor(a, 2);
            if (part == 0) ys[q] = a;
        }
    }
    __syncthreads();
    if (rb < 0) {
        for (int q = tid; q < k; q += 256) {
            const int pt = V.ptype[c0 + q];
            double z;
            if (pt == 1) z = ys[q] * V.dinv[c0 + q];
            else if (pt == 2) z = V.dinv[c0 + q] * ys[q] + V.doff[c0 + q] * ys[q + 1];
            else z = V.doff[c0 + q - 1] * ys[q - 1] + V.dinv[c0 + q] * ys[q];
            V.zb[c0 + q] = z;
        }
        return;
    }
    const int i = k + rb * 64 + lane;
    const bool ok = i < m;
    const double* Lg = V.L + M.panel_off + (ok ? i : k);
    double t0 = 0.0, t1 = 0.0;
    for (int p = wave * 8; p < k; p += 32) {
        double l[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) l[u] = (p + u < k) ? Lg[(size_t)(p + u) * M.ldp] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; u += 2) { t0 += l[u] * ((p + u < k) ? ys[p + u] : 0.0); t1 += l[u + 1] * ((p + u + 1 < k) ? ys[p + u + 1] : 0.0); }
    }
    red[wave][lane] = t0 + t1;
    __syncthreads();
    if (wave == 0 && ok) {
        const double t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        cv[i] = (M.solo == 1 ? cv[i] : 0.0) - t;
    }
}

// k <= 64 variant of k_fwd_solo with every load that only depends on the front record issued up front (the pivot order, the
// inverse and the panel entries do not depend on the right-hand side): one dependent round trip instead of four.  On this part a
// dependent global access costs 1.5-2 us, which is what a launch of these latency-bound kernels is made of.
__global__ __launch_bounds__(256) void k_fwd_solo64(DevView V, int list_off)
{
    __shared__ double bp[64], ys[64];
    __shared__ double red[4][64];
    const FrontMeta M = V.fmeta[list_off + blockIdx.y];
    const int k = M.k, m = M.m, c0 = M.c0;
    const int rb = (int)blockIdx.x - 1;                       // -1: pivot part
    if (rb >= 0 && k + rb * 64 >= m) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* cv = V.cvec + M.cv;
    // ---- everything that depends on M only ----
    const int lpv = (tid < k) ? V.lperm[c0 + tid] : 0;
    const double* Mg = V.minv + M.minv_off;
    const int q = tid >> 2, part = tid & 3;
    double mreg[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { const int p = part + 4 * u; mreg[u] = (q < k && p <= q) ? Mg[q + (size_t)p * k] : 0.0; }
    const int i = k + rb * 64 + lane;
    const bool ok = rb >= 0 && i < m;
    const double* Lg = V.L + M.panel_off + (ok ? i : k);
    double lreg[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { const int p = wave * 8 + (u & 7) + 32 * (u >> 3); lreg[u] = (ok && p < k) ? Lg[(size_t)p * M.ldp] : 0.0; }
    const double cvi = (ok && wave == 0 && M.solo == 1) ? cv[i] : 0.0;
    int pt = 1; double dq = 0.0, oq = 0.0, oq1 = 0.0;
    if (rb < 0 && tid < k) { pt = V.ptype[c0 + tid]; dq = V.dinv[c0 + tid]; oq = V.doff[c0 + tid]; oq1 = tid > 0 ? V.doff[c0 + tid - 1] : 0.0; }
    // ---- dependent on the pivot order ----
    if (tid < 64) bp[tid] = tid < k ? V.xw[c0 + lpv] + (M.solo == 1 ? cv[lpv] : 0.0) : 0.0;      // (ALL 64 entries: the product below runs over them with zeros of mreg, and 0 x a NaN left in LDS by an earlier kernel is NaN)
    __syncthreads();
    {
        double a = 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) a += mreg[u] * bp[(part + 4 * u) & 63];
        a += __shfl_xor(a, 1); a += __shfl_xor(a, 2);
        if (part == 0 && q < k) ys[q] = a;
    }
    __syncthreads();
    if (rb < 0) {
        if (tid < k) {
            double z;
            if (pt == 1) z = ys[tid] * dq;
            else if (pt == 2) z = dq * ys[tid] + oq * ys[tid + 1];
            else z = oq1 * ys[tid - 1] + dq * ys[tid];
            V.zb[c0 + tid] = z;
        }
        return;
    }
    double t0 = 0.0, t1 = 0.0;
#pragma unroll
    for (int u = 0; u < 16; u += 2) {
        const int p = wave * 8 + (u & 7) + 32 * (u >> 3);
        t0 += lreg[u] * ((p < k) ? ys[p] : 0.0); t1 += lreg[u + 1] * ((p + 1 < k) ? ys[p + 1] : 0.0);
    }
    red[wave][lane] = t0 + t1;
    __syncthreads();
    if (wave == 0 && ok) cv[i] = cvi - ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
}

__global__ __launch_bounds__(256) void k_fwd_grp_upd(DevView V, int list_off)
{
    // 64 rows per workgroup (one per lane); the group's columns are dealt to the 4 wavefronts in blocks of 8 (8 coalesced
    // loads in flight per lane), the 4 partial sums are combined through LDS in fixed order
    __shared__ double ys[256];
    __shared__ double red[4][64];
    const FrontMeta M = V.fmeta[list_off + blockIdx.y];
    if (M.grem != 0) return;
    if (M.k + blockIdx.x * 64 >= M.m) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nl = M.gpos + 1;
    int cb = 0;
    for (int j = 0; j < nl; ++j) {
        const GroupLink G = V.gtab[M.gbase + j];
        for (int p = tid; p < G.k; p += 256) ys[cb + p] = V.ybuf[G.c0 + p];
        cb += G.k;
    }
    __syncthreads();
    const int i = M.k + blockIdx.x * 64 + lane;
    const bool ok = i < M.m;
    double t0 = 0.0, t1 = 0.0;
    cb = 0;
    for (int j = 0; j < nl; ++j) {
        const GroupLink G = V.gtab[M.gbase + j];
        const double* Lg = V.L + G.panel_off + (G.m - M.m) + (ok ? i : M.k);
        const double* y = ys + cb;
        const int k = G.k;
        for (int p = wave * 8; p < k; p += 32) {
            double l[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) l[u] = (p + u < k) ? Lg[(size_t)(p + u) * G.ldp] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; u += 2) { t0 += l[u] * ((p + u < k) ? y[p + u] : 0.0); t1 += l[u + 1] * ((p + u + 1 < k) ? y[p + u + 1] : 0.0); }
        }
        cb += k;
    }
    red[wave][lane] = t0 + t1;
    __syncthreads();
    if (wave == 0 && ok) V.cvec[M.cv + i] -= (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

__global__ __launch_bounds__(256) void k_bwd_grp_dot(DevView V, int list_off)
{
    // 256 rows per workgroup (4 per lane); column blocks of 4 are dealt to the wavefronts: 16 loads in flight per lane
    __shared__ double xs[256];
    const FrontMeta M = V.fmeta[list_off + blockIdx.y];
    if (M.grem != 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ibase = M.k + blockIdx.x * 256;
    if (ibase >= M.m) return;
    const int nrow = min(256, M.m - ibase);
    xs[tid] = (tid < nrow) ? V.xw[V.sn_rows[M.r0 + ibase + tid]] : 0.0;
    __syncthreads();
    double* part = V.gpart + M.gpart + (size_t)blockIdx.x * M.gcols;
    const int nl = M.gpos + 1;
    const double x0 = xs[lane], x1 = xs[lane + 64], x2 = xs[lane + 128], x3 = xs[lane + 192];
    const bool v0 = lane < nrow, v1 = lane + 64 < nrow, v2 = lane + 128 < nrow, v3 = lane + 192 < nrow;
    int cb = 0;
    for (int j = 0; j < nl; ++j) {
        const GroupLink G = V.gtab[M.gbase + j];
        const double* Lg = V.L + G.panel_off + (G.m - M.m) + ibase + lane;
        const int k = G.k;
        for (int pb = wave * 4; pb < k; pb += 16) {
            double t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool cv = pb + u < k;
                const double* c = Lg + (size_t)(pb + u) * G.ldp;
                const double a0 = (cv && v0) ? c[0] : 0.0, a1 = (cv && v1) ? c[64] : 0.0, a2 = (cv && v2) ? c[128] : 0.0, a3 = (cv && v3) ? c[192] : 0.0;
                t[u] = (a0 * x0 + a1 * x1) + (a2 * x2 + a3 * x3);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { t[u] = wave_sum(t[u]); if (lane == 0 && pb + u < k) part[cb + pb + u] = t[u]; }
        }
        cb += k;
    }
}

// per-link (one link per solve unit), k <= 64 variants of k_bwd_grp_dot / k_bwd_grp with the loads that only depend on the front
// record hoisted to the top (same arithmetic, same order of summation)
__global__ __launch_bounds__(256) void k_bwd_dot64(DevView V, int list_off)
{
    const FrontMeta M = V.fmeta[list_off + blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ibase = M.k + blockIdx.x * 256;
    if (ibase >= M.m) return;
    const int nrow = min(256, M.m - ibase), k = M.k;
    const bool v0 = lane < nrow, v1 = lane + 64 < nrow, v2 = lane + 128 < nrow, v3 = lane + 192 < nrow;
    // the panel entries first (they depend on the front record only) ...
    const double* Lg = V.L + M.panel_off + ibase + lane;
    double lv[4][4][4];                                      // [pass][column of the pass][row strip]
#pragma unroll
    for (int ps = 0; ps < 4; ++ps)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int col = wave * 4 + 16 * ps + u;
            const bool cv = col < k;
            const double* c = Lg + (size_t)col * M.ldp;
            lv[ps][u][0] = (cv && v0) ? c[0] : 0.0; lv[ps][u][1] = (cv && v1) ? c[64] : 0.0;
            lv[ps][u][2] = (cv && v2) ? c[128] : 0.0; lv[ps][u][3] = (cv && v3) ? c[192] : 0.0;
        }
    // ... then the two dependent hops to the solution entries of the rows
    const int r0i = M.r0 + ibase;
    const double x0 = v0 ? V.xw[V.sn_rows[r0i + lane]] : 0.0, x1 = v1 ? V.xw[V.sn_rows[r0i + lane + 64]] : 0.0;
    const double x2 = v2 ? V.xw[V.sn_rows[r0i + lane + 128]] : 0.0, x3 = v3 ? V.xw[V.sn_rows[r0i + lane + 192]] : 0.0;
    double* part = V.gpart + M.gpart + (size_t)blockIdx.x * M.gcols;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int col = wave * 4 + 16 * ps + u;
            double t = (lv[ps][u][0] * x0 + lv[ps][u][1] * x1) + (lv[ps][u][2] * x2 + lv[ps][u][3] * x3);
            t = wave_sum(t);
            if (lane == 0 && col < k) part[col] = t;
        }
}
__global__ __launch_bounds__(256) void k_bwd_fin64(DevView V, int list_off)
{
    __shared__ double ws[64];
    const FrontMeta M = V.fmeta[list_off + blockIdx.x];
    const int tid = threadIdx.x;
    const int k = M.k, c0 = M.c0;
    const int nch = (M.m - M.k + 255) / 256;
    const double* part = V.gpart + M.gpart;
    const double* Mg = V.minv + M.minv_off;
    const int pcol = tid >> 2, part4 = tid & 3;
    double mreg[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { const int q = pcol + part4 + 4 * u; mreg[u] = (pcol < k && q < k) ? Mg[q + (size_t)pcol * k] : 0.0; }
    const int lpv = (part4 == 0 && pcol < k) ? V.lperm[c0 + pcol] : 0;
    if (tid < 64) ws[tid] = 0.0;
    __syncthreads();
    if (tid < k) {
        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
        int c = 0;
        for (; c + 3 < nch; c += 4) { t0 += part[(size_t)c * M.gcols + tid]; t1 += part[(size_t)(c + 1) * M.gcols + tid];
                                      t2 += part[(size_t)(c + 2) * M.gcols + tid]; t3 += part[(size_t)(c + 3) * M.gcols + tid]; }
        for (; c < nch; ++c) t0 += part[(size_t)c * M.gcols + tid];
        ws[tid] = V.zb[c0 + tid] - ((t0 + t1) + (t2 + t3));
    }
    __syncthreads();
    double a = 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) a += mreg[u] * ws[(pcol + part4 + 4 * u) & 63];       // (entries beyond k are zero in mreg)
    a += __shfl_xor(a, 1); a += __shfl_xor(a, 2);
    if (part4 == 0 && pcol < k) V.xw[c0 + lpv] = a;
}

__global__ __launch_bounds__(256) void k_bwd_grp(DevView V, int list_off)
{
    __shared__ double ws[128], xs[256];
    const FrontMeta M = V.fmeta[list_off + blockIdx.x];
    if (M.grem != 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nch = (M.m - M.k + 255) / 256;
    const double* part = V.gpart + M.gpart;
    int cb = M.gcols;
    for (int j = M.gpos; j >= 0; --j) {
        const GroupLink G = V.gtab[M.gbase + j];
        const int k = G.k, c0 = G.c0;
        cb -= k;                                           // first group column of this link
        const int rem = M.gcols - cb - k;                  // pivots of the later links = this link's first update rows
        for (int p = tid; p < k; p += 256) {
            double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
            int c = 0;
            for (; c + 3 < nch; c += 4) { t0 += part[(size_t)c * M.gcols + cb + p]; t1 += part[(size_t)(c + 1) * M.gcols + cb + p];
                                          t2 += part[(size_t)(c + 2) * M.gcols + cb + p]; t3 += part[(size_t)(c + 3) * M.gcols + cb + p]; }
            for (; c < nch; ++c) t0 += part[(size_t)c * M.gcols + cb + p];
            ws[p] = V.zb[c0 + p] - ((t0 + t1) + (t2 + t3));
        }
        for (int i = tid; i < rem; i += 256) xs[i] = V.xw[V.sn_rows[G.r0 + k + i]];
        __syncthreads();
        if (rem > 0) {
            const double* Lg = V.L + G.panel_off + k;
            for (int pb = wave * 4; pb < k; pb += 16) {           // 4 columns per pass: their loads are all in flight together
                double t[4] = {0.0, 0.0, 0.0, 0.0};
                for (int i = lane; i < rem; i += 64) {
                    const double x = xs[i];
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (pb + u < k) t[u] += Lg[i + (size_t)(pb + u) * G.ldp] * x;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { t[u] = wave_sum(t[u]); if (lane == 0 && pb + u < k) ws[pb + u] -= t[u]; }
            }
            __syncthreads();
        }
        const double* Mg = V.minv + G.minv_off;
        {   // x_p = sum_{q >= p} Minv(q,p) w_q : 4 threads per column (contiguous quarter-interleaved walk), combined by shuffles
            const int part = tid & 3;
            for (int p = tid >> 2; p < k; p += 64) {
                double a = 0.0;
                int q = p + part;
                for (; q + 12 < k; q += 16) {
                    const double m0 = Mg[q + (size_t)p * k], m1 = Mg[q + 4 + (size_t)p * k], m2 = Mg[q + 8 + (size_t)p * k], m3 = Mg[q + 12 + (size_t)p * k];
                    a += m0 * ws[q] + m1 * ws[q + 4] + m2 * ws[q + 8] + m3 * ws[q + 12];
                }
                for (; q < k; q += 4) a += Mg[q + (size_t)p * k] * ws[q];
                a += __shfl_xor(a, 1); a += __shfl_xor(a, 2);
                if (part == 0) V.xw[c0 + V.lperm[c0 + p]] = a;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// sync-free chain sweeps (see ChainLink / ChainDesc)
// ------------------------------------------------------------------------------------------------
__global__ void k_bump_epoch(int* e) { if (threadIdx.x == 0 && blockIdx.x == 0) *e += 1; }
template <int NAP>
__device__ __forceinline__ void chain_wait(const int* flag, const int epoch, int* err)
{
    if (threadIdx.x == 0) {
        // bounded: a producer that never shows up (it cannot: it has a lower workgroup index, so it was dispatched before) must not hang the GPU
        // NAP: 64-cycle units between two polls -- long where the flag is not the last one the workgroup waits for (fewer requests in the
        // L2 queues the critical hop goes through)
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
            __builtin_amdgcn_s_sleep(NAP);
            if (++spins > (1 << 24)) { *err = 1; break; }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
__device__ __forceinline__ void chain_signal(int* flag, const int epoch)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // every wavefront publishes its own stores
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ---- the messages of the sweeps ----
// The 8 XCDs of the part have an L2 each; what one workgroup stores becomes visible to another XCD only through the fabric.  A release
// fence at agent scope writes the WHOLE L2 back (buffer_wbl2) and an acquire invalidates it: measured 1.6-2.1 us per one-way message of
// 64 doubles with "data, fence, flag / poll, fence, load" (tools/micro/pingpong.hip).  So nothing here uses fences:
//   * a link's solution travels as 64 {value, tag} pairs, each ONE 16-byte agent-coherent store (sc1: through the L2 to the fabric) that the
//     consumers poll with 16-byte agent-coherent loads -- value and tag arrive together, no flag, no fence: 0.4-0.5 us per message;
//   * everything else a workgroup hands to another one inside a launch (rows of a chain vector, solution entries) is stored and loaded
//     agent-coherently too, and the flag that announces it goes out after the stores have been acknowledged (s_waitcnt vmcnt(0)).
__device__ __forceinline__ v2d ld_tag(const v2d* p) { v2d r; asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory"); return r; }
__device__ __forceinline__ void st_tag(v2d* p, v2d v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ double ld_coh(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_coh(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a value every lane holds identically, moved to scalar registers (the compiler cannot see that what a vector load of a per-workgroup record returns is uniform)
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ long long uni(long long x) { return ((long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(x >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ ChainLink uni(const ChainLink& a)
{
    ChainLink r; r.panel_off = uni(a.panel_off); r.minv_off = uni(a.minv_off); r.c0 = uni(a.c0); r.k = uni(a.k); r.ldp = uni(a.ldp); r.s = uni(a.s); r.r0 = uni(a.r0); r.koff = uni(a.koff); r.fi = uni(a.fi); r.pad1 = 0;
    return r;
}
__device__ __forceinline__ ChainDesc uni(const ChainDesc& a)
{
    ChainDesc r; r.cvb = uni(a.cvb); r.link0 = uni(a.link0); r.nlinks = uni(a.nlinks); r.tail = uni(a.tail); r.ktot = uni(a.ktot); r.wg0f = uni(a.wg0f); r.wg0b = uni(a.wg0b);
    r.ch0 = uni(a.ch0); r.ch1 = uni(a.ch1); r.alias0 = uni(a.alias0); r.s0 = uni(a.s0); r.init = uni(a.init); r.gw0 = uni(a.gw0); r.gw1 = uni(a.gw1); r.tf0 = uni(a.tf0);
    r.pw0 = uni(a.pw0); r.pw1 = uni(a.pw1); r.dot0 = uni(a.dot0); r.pad0 = 0;
    return r;
}
// the whole wavefront waits until the (first k of the) 64 tagged entries at p carry this solve's tag; returns lane's value (0 beyond k).
// Dozens of workgroups wait for the same message, and every poll is a request to the ONE memory channel that holds it: polls of all 64
// entries from every waiter queue up there and the message the next link is waiting for arrives 1.7 us late instead of 0.45 (measured).  So a
// waiter polls the first entry only (one request per wavefront), and the further it is from needing the message -- dist: hops between this
// message and the last one it waits for -- the longer it sleeps between polls; then one load of all entries (repeated if the producer's other
// wavefronts have not landed yet).  There is ALWAYS an s_sleep between two polls: re-issued back to back (~100 ns apart) by the same wavefront,
// the agent-coherent load of the same line kept returning the value of the first poll -- for seconds -- on this part.
__device__ __forceinline__ double tag_await(const v2d* p, const int lane, const int k, const double ep, const int dist, int* err)
{
    int spins = 0;
    const int ehi = __double2hiint(ep), elo = __double2loint(ep);
    while (dist > 0) {          // (the message the workgroup needs next is polled in full straight away: one round trip less)
        const v2d h = ld_tag(p);
        if (__builtin_amdgcn_readfirstlane(__double2hiint(h.y)) == ehi && __builtin_amdgcn_readfirstlane(__double2loint(h.y)) == elo) break;      // (scalar branch)
        __builtin_amdgcn_s_sleep(1);
        for (int d = 0; d < min(dist, 6); ++d) __builtin_amdgcn_s_sleep(10);
        if (++spins > (1 << 22)) { *err = 1; break; }       // (cannot happen: the producer has a lower workgroup index, it was dispatched first)
    }
    const v2d* q = p + (lane < k ? lane : 0);
    v2d v;
    while (true) {
        v = ld_tag(q);
        if (__all(v.y == ep)) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) { *err = 1; break; }
    }
    return lane < k ? v.x : 0.0;
}
// flags of the sweeps (FLAG_STRIDE ints apart): raised after this workgroup's coherent stores are acknowledged; awaited side by side, one per lane
__device__ __forceinline__ void flag_raise(int* flag, const int epoch)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void flags_await(const int* flags, const int* list, const int f0, const int f1, const int epoch, int* err)
{
    if (threadIdx.x < 64)
        for (int f = f0 + (int)threadIdx.x; f < f1; f += 64) {
            const int* flag = flags + (size_t)(list ? list[f] : f) * FLAG_STRIDE;
            int spins = 0;
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > (1 << 22)) { *err = 1; break; }
            }
        }
    __syncthreads();
}

// Forward sweep of a segment.  Workgroup = one link of a chain (64 pivot rows of the chain vector) or 64 rows beyond the chain; it OWNS its rows:
// their running value sits in registers (4 lanes per row, each with 16 of the 64 columns of a panel block) until every earlier link of the
// chain has been applied.  Five wavefronts:
//   * wavefront 4 is the POLLER: it waits for the tagged y of one earlier link after the other and puts it into LDS -- it has no other memory
//     operation in flight, ever: loads return to a wavefront in order, so a poll issued behind the prefetch of a panel block (HBM, ~2 us) would
//     see the message that much late (measured: 2.4 us per hop that way, 0.9 us for the bare message, tools/micro/chain_hop.hip);
//   * wavefronts 0-3 apply a message as soon as the barrier says it is there: 16 FMAs per lane, two DPP adds.  The panel blocks of the NEXT TWO
//     links are in registers or on their way (a block is requested two hops before it is needed).
// A link then permutes (LDS, one barrier), applies its stored inverse from registers and publishes y.  Everything that does not depend on the
// incoming vectors -- inverse, pivot data, panel blocks, the children's inverse row maps -- is requested before the first wait.
__global__ __launch_bounds__(320) void k_fwd_chain(DevView V, int wg0)
{
    __shared__ double ymsg[2][64], bps[64], yss[64];
    __shared__ int ipos[64], lc0[64], lkk[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = (tid >> 2) & 63, part = tid & 3;
    const bool poller = wave == 4;
    const int di = V.chwg_f[wg0 + (int)blockIdx.x];
    const ChainDesc C = uni(V.chdesc[di]);
    unsigned long long* tr = V.strace ? V.strace + 4 * (size_t)(wg0 + (int)blockIdx.x) : nullptr;
    if (tr && tid == 0) tr[0] = wall_clock64();
    const int w = (int)blockIdx.x - C.wg0f;
    const int epoch = uni(*V.sepoch);
    const double ep = (double)epoch;
    int* err = V.sepoch + 1;
    const bool is_link = w < C.nlinks;
    const ChainLink Me = uni(V.chlink[C.link0 + (is_link ? w : C.nlinks - 1)]);
    const int roff = is_link ? Me.koff : C.ktot + 64 * (w - C.nlinks);       // my rows inside the chain vector
    const int rows = is_link ? Me.k : min(64, C.tail - 64 * (w - C.nlinks));
    const bool rok = !poller && row < rows;
    double* cvp = V.cvec + C.cvb;
    const int k = Me.k, c0 = Me.c0;
    const int nprev = is_link ? w : C.nlinks;
    double mreg[16], lrA[16], lrB[16];
    int pt = 1; double dq = 0.0, oq = 0.0, oq1 = 0.0;
    auto fetch_block = [&](double (&lr)[16], const int i) {      // my rows of link i's panel
        const ChainLink L = V.chlink[C.link0 + i];
        const double* Lb = V.L + uni(L.panel_off) + (roff - uni(L.koff));      // (uniform base + 32-bit lane offsets: one address register pair, not 16)
        const int ldl = uni(L.ldp), kl = uni(L.k);
        const int off = (rok ? row : 0) + part * ldl;
#pragma unroll
        for (int u = 0; u < 16; ++u) lr[u] = (rok && part + 4 * u < kl) ? Lb[off + 4 * u * ldl] : 0.0;
    };
    double xb = 0.0;
    if (!poller) {
        if (is_link) {
            const double* Mg = V.minv + Me.minv_off;
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int pp = part + 4 * u; mreg[u] = (row < k && pp <= row) ? Mg[row + (size_t)pp * k] : 0.0; }      // Minv(row, pp)
            if (tid < 64) bps[tid] = 0.0;          // (entries beyond k meet zeros of mreg below: they must not be whatever an earlier kernel left in LDS -- 0 x NaN)
            if (tid < k) ipos[V.lperm[c0 + tid]] = tid;
            if (part == 0 && row < k) { pt = V.ptype[c0 + row]; dq = V.dinv[c0 + row]; oq = V.doff[c0 + row]; oq1 = row > 0 ? V.doff[c0 + row - 1] : 0.0; }
            if (rok) xb = V.xw[c0 + row];
        }
        if (nprev > 0) fetch_block(lrA, 0);
        if (nprev > 1) fetch_block(lrB, 1);
    } else {
        for (int i = lane; i < min(nprev, 64); i += 64) { const ChainLink L = V.chlink[C.link0 + i]; lc0[i] = L.c0; lkk[i] = L.k; }
    }
    // ---- the value my rows start from ----
    double acc = 0.0;
    if (C.init == 2) {
        // rows of the children's fronts that land on mine (inverse row maps: static), then the children's chains must be complete
        int inv[4]; long long cvb[4]; int nc = 0, cp = C.ch0;
        if (!poller)
            for (; cp < C.ch1 && nc < 4; ++cp) {
                const ChildMeta Cm = V.cmeta[cp];
                if (Cm.aliased) continue;
                inv[nc] = rok ? V.relinv[Cm.inv + roff + row] : -1; cvb[nc] = Cm.cvbase; ++nc;
            }
        flags_await(V.sflag_t, V.chwait, C.gw0, C.gw1, epoch, err);
        if (!poller) {
            double t = (C.alias0 && rok) ? ld_coh(&cvp[roff + row]) : 0.0;
            for (int c = 0; c < nc; ++c) if (inv[c] >= 0) t += ld_coh(V.cvec + cvb[c] + inv[c]);
            for (; cp < C.ch1; ++cp) {          // (more than 4 gathered children: not a nested-dissection tree)
                const ChildMeta Cm = V.cmeta[cp];
                if (Cm.aliased) continue;
                const int iv = rok ? V.relinv[Cm.inv + roff + row] : -1;
                if (iv >= 0) t += ld_coh(V.cvec + Cm.cvbase + iv);
            }
            acc = xb + t;
        }
    } else {
        __syncthreads();                    // (ipos)
        if (!poller) acc = (C.init == 0 && rok) ? xb + cvp[roff + row] : xb;
    }
    const int myipos = (!poller && is_link && row < k) ? ipos[row] : 0;
    if (tr && tid == 0) tr[1] = wall_clock64();
    auto poll = [&](const int i, double* dst) {
        if ((i & 63) == 0 && i > 0) { for (int q = lane; q < min(nprev - i, 64); q += 64) { const ChainLink L = V.chlink[C.link0 + i + q]; lc0[q] = L.c0; lkk[q] = L.k; } __builtin_amdgcn_wave_barrier(); }
        dst[lane] = tag_await(V.ytag + lc0[i & 63], lane, lkk[i & 63], ep, nprev - 1 - i, err);
    };
    auto step = [&](double (&lr)[16], const double* ym, const int i) {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int u = 0; u < 16; u += 2) { a0 += lr[u] * ym[part + 4 * u]; a1 += lr[u + 1] * ym[part + 4 * u + 4]; }
        if (i + 2 < nprev) fetch_block(lr, i + 2);          // two hops ahead
        double t = a0 + a1;
        t += dpp_f64<0xB1>(t); t += dpp_f64<0x4E>(t);       // the 4 lanes of a row are a DPP quad
        acc -= t;
    };
    for (int i = 0; i < nprev; i += 2) {
        if (poller) poll(i, ymsg[0]);
        __syncthreads();
        if (!poller) step(lrA, ymsg[0], i);
        if (i + 1 < nprev) {
            if (poller) poll(i + 1, ymsg[1]);
            __syncthreads();
            if (!poller) step(lrB, ymsg[1], i + 1);
        }
    }
    if (poller) return;
    if (tr && tid == 0) tr[2] = wall_clock64();
    if (!is_link) {          // rows beyond the chain: complete, the parent's chain may take them
        if (rok && part == 0) st_coh(&cvp[roff + row], acc);
        flag_raise(&V.sflag_t[(size_t)(C.tf0 + (w - C.nlinks)) * FLAG_STRIDE], epoch);
        if (tr && tid == 0) tr[3] = wall_clock64();
        return;
    }
    if (part == 0 && row < k) bps[myipos] = acc;              // P b
    __syncthreads();
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int u = 0; u < 16; u += 2) { a0 += mreg[u] * bps[part + 4 * u]; a1 += mreg[u + 1] * bps[part + 4 * u + 4]; }      // (mreg is zero beyond the row)
    double y = a0 + a1;
    y += dpp_f64<0xB1>(y); y += dpp_f64<0x4E>(y);
    if (part == 0 && row < k) { v2d m; m.x = y; m.y = ep; st_tag(V.ytag + c0 + row, m); }
    if (tr && tid == 0) tr[3] = wall_clock64();
    // off the chain's critical path: z = D^{-1} y for the backward sweep
    if (part == 0) yss[row] = y;
    __syncthreads();
    if (part == 0 && row < k) {
        double z;
        if (pt == 1) z = y * dq;
        else if (pt == 2) z = dq * y + oq * yss[row + 1];
        else z = oq1 * yss[row - 1] + dq * y;
        V.ybuf[c0 + row] = y;
        V.zb[c0 + row] = z;
    }
}
// Backward sweep of a segment, the parents' chains first.  Per chain, two kinds of workgroups:
//   * DOT workgroups, one per (link, 256 rows beyond the chain): the 256 x 64 block of the link's panel goes into registers BEFORE anything is awaited
//     (it does not depend on the solution), then -- once every link of the parent's chain has raised its flag -- the rows' solution entries, 64 FMAs per
//     lane, a DPP sum per column, 64 partial sums stored coherently, flag.  The streaming of a chain's panels is spread over the whole machine and is
//     over when the parent finishes; up to round 2 one workgroup per link streamed its own rows after the wait (18-30 us per tree level);
//   * LINK workgroups, top link first, walk the chain as in the forward sweep: wavefront 4 polls the tagged x of the later links, wavefronts 0-3 apply the
//     64 x 64 block of MY panel that meets a later link's rows (16 FMAs per lane, two DPP adds; blocks requested two hops ahead); finally L11^{-T} from
//     registers, x published tagged (for the chain) and plain (for everything below), flag for the children's chains.
__global__ __launch_bounds__(320) void k_bwd_chain(DevView V, int wg0)
{
    __shared__ double ws[64], xmsg[2][64];
    __shared__ int lc0[64], lkk[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = (tid >> 2) & 63, part = tid & 3;
    const bool poller = wave == 4;
    const ChainDesc C = uni(V.chdesc[V.chwg_b[wg0 + (int)blockIdx.x]]);
    unsigned long long* tr = V.strace ? V.strace + 4 * (size_t)(V.strace_b + wg0 + (int)blockIdx.x) : nullptr;
    if (tr && tid == 0) tr[0] = wall_clock64();
    const int r = (int)blockIdx.x - C.wg0b;
    const int nbk = (C.tail + 255) >> 8, ndots = C.nlinks * nbk;
    const int epoch = uni(*V.sepoch);
    const double ep = (double)epoch;
    int* err = V.sepoch + 1;
    if (r < ndots) {
        // ---------------- dot workgroup ----------------
        if (poller) return;
        const int jj = r / nbk, b = r - jj * nbk;                // (jj = 0: the top link)
        const ChainLink Me = uni(V.chlink[C.link0 + C.nlinks - 1 - jj]);
        const int k = Me.k, toff = C.ktot - Me.koff, ibase = b * 256, nrow = min(256, C.tail - ibase);
        const bool v0 = lane < nrow, v1 = lane + 64 < nrow, v2 = lane + 128 < nrow, v3 = lane + 192 < nrow;
        const double* Lg = V.L + Me.panel_off + toff + ibase + lane;
        double lv[4][4][4];                                      // [pass][column of the pass][row strip]
#pragma unroll
        for (int ps = 0; ps < 4; ++ps)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cc = wave * 4 + 16 * ps + u;
                const bool cv = cc < k;
                const double* c = Lg + (size_t)(cv ? cc : 0) * Me.ldp;
                lv[ps][u][0] = (cv && v0) ? c[0] : 0.0; lv[ps][u][1] = (cv && v1) ? c[64] : 0.0;
                lv[ps][u][2] = (cv && v2) ? c[128] : 0.0; lv[ps][u][3] = (cv && v3) ? c[192] : 0.0;
            }
        const int r0i = Me.r0 + toff + ibase + lane;
        const int i0 = v0 ? V.sn_rows[r0i] : 0, i1 = v1 ? V.sn_rows[r0i + 64] : 0, i2 = v2 ? V.sn_rows[r0i + 128] : 0, i3 = v3 ? V.sn_rows[r0i + 192] : 0;
        flags_await(V.sflag_b, V.chwait, C.pw0, C.pw1, epoch, err);
        if (tr && tid == 0) tr[1] = wall_clock64();
        const double x0 = v0 ? ld_coh(&V.xw[i0]) : 0.0, x1 = v1 ? ld_coh(&V.xw[i1]) : 0.0, x2 = v2 ? ld_coh(&V.xw[i2]) : 0.0, x3 = v3 ? ld_coh(&V.xw[i3]) : 0.0;
        double* dp = V.dpart + (size_t)(C.dot0 + r) * 64;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cc = wave * 4 + 16 * ps + u;
                double t = (lv[ps][u][0] * x0 + lv[ps][u][1] * x1) + (lv[ps][u][2] * x2 + lv[ps][u][3] * x3);
                t = wave_sum_dpp(t);
                if (lane == 0 && cc < k) st_coh(&dp[cc], t);
            }
        if (tr && tid == 0) tr[2] = wall_clock64();
        flag_raise(&V.sflag_dot[(size_t)(C.dot0 + r) * FLAG_STRIDE], epoch);
        if (tr && tid == 0) tr[3] = wall_clock64();
        return;
    }
    // ---------------- link workgroup ----------------
    const int jt = r - ndots;                                  // 0: the top link, the head of the dependency chain
    const int j = C.nlinks - 1 - jt;
    const int nlater = jt;                                     // links whose x I wait for, top first: link nlinks - 1 - i is message i
    const ChainLink Me = uni(V.chlink[C.link0 + j]);
    const int k = Me.k, c0 = Me.c0;
    double mreg[16], lrA[16], lrB[16];
    int lpv = 0;
    auto fetch_block = [&](double (&lr)[16], const int i) {      // rows of the pivots of link nlinks - 1 - i in my panel, my column; lane part has rows part, part + 4, ...
        const ChainLink L = V.chlink[C.link0 + C.nlinks - 1 - i];
        const double* Lb = V.L + uni(Me.panel_off) + (uni(L.koff) - uni(Me.koff));
        const int kl = uni(L.k);
        const int off = (col < k ? col : 0) * uni(Me.ldp) + part;
#pragma unroll
        for (int u = 0; u < 16; ++u) lr[u] = (col < k && part + 4 * u < kl) ? Lb[off + 4 * u] : 0.0;
    };
    if (!poller) {
        const double* Mg = V.minv + Me.minv_off;
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int q = col + part + 4 * u; mreg[u] = (col < k && q < k) ? Mg[q + (size_t)col * k] : 0.0; }      // Minv(q, col), q >= col
        if (col < k) lpv = V.lperm[c0 + col];
        if (nlater > 0) fetch_block(lrA, 0);
        if (nlater > 1) fetch_block(lrB, 1);
    } else {
        for (int i = lane; i < min(nlater, 64); i += 64) { const ChainLink L = V.chlink[C.link0 + C.nlinks - 1 - i]; lc0[i] = L.c0; lkk[i] = L.k; }
    }
    const double zv = (tid < k) ? V.zb[c0 + tid] : 0.0;
    if (C.tail > 0) {
        // the partial sums of my dot workgroups, added in block order
        flags_await(V.sflag_dot, nullptr, C.dot0 + jt * nbk, C.dot0 + (jt + 1) * nbk, epoch, err);
        if (tr && tid == 0) tr[1] = wall_clock64();
        if (tid < 64) {
            double t = 0.0;
            const double* dp = V.dpart + (size_t)(C.dot0 + jt * nbk) * 64 + tid;
            for (int b = 0; b < nbk; ++b) t += (tid < k) ? ld_coh(dp + (size_t)b * 64) : 0.0;
            ws[tid] = zv - t;
        }
    } else if (tid < 64) ws[tid] = zv;
    __syncthreads();
    double wv = poller ? 0.0 : ws[col];
    __syncthreads();                                           // (everybody has read ws)
    if (tr && tid == 0) tr[2] = wall_clock64();
    auto poll = [&](const int i, double* dst) {
        if ((i & 63) == 0 && i > 0) { for (int q = lane; q < min(nlater - i, 64); q += 64) { const ChainLink L = V.chlink[C.link0 + C.nlinks - 1 - i - q]; lc0[q] = L.c0; lkk[q] = L.k; } __builtin_amdgcn_wave_barrier(); }
        dst[lane] = tag_await(V.xtag + lc0[i & 63], lane, lkk[i & 63], ep, nlater - 1 - i, err);
    };
    auto step = [&](double (&lr)[16], const double* xm, const int i) {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int u = 0; u < 16; u += 2) { a0 += lr[u] * xm[part + 4 * u]; a1 += lr[u + 1] * xm[part + 4 * u + 4]; }
        if (i + 2 < nlater) fetch_block(lr, i + 2);
        double t = a0 + a1;
        t += dpp_f64<0xB1>(t); t += dpp_f64<0x4E>(t);
        wv -= t;
    };
    for (int i = 0; i < nlater; i += 2) {
        if (poller) poll(i, xmsg[0]);
        __syncthreads();
        if (!poller) step(lrA, xmsg[0], i);
        if (i + 1 < nlater) {
            if (poller) poll(i + 1, xmsg[1]);
            __syncthreads();
            if (!poller) step(lrB, xmsg[1], i + 1);
        }
    }
    if (poller) return;
    if (part == 0) ws[col] = wv;
    __syncthreads();
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int u = 0; u < 16; u += 2) { a0 += mreg[u] * ws[(col + part + 4 * u) & 63]; a1 += mreg[u + 1] * ws[(col + part + 4 * u + 4) & 63]; }     // (mreg is zero beyond k)
    double x = a0 + a1;
    x += dpp_f64<0xB1>(x); x += dpp_f64<0x4E>(x);
    if (part == 0 && col < k) {
        v2d m; m.x = x; m.y = ep;
        st_tag(V.xtag + c0 + lpv, m);
        st_coh(&V.xw[c0 + lpv], x);
    }
    flag_raise(&V.sflag_b[(size_t)Me.fi * FLAG_STRIDE], epoch);
    if (tr && tid == 0) tr[3] = wall_clock64();
}

// ================================================================================================
// BIG fronts (order > 128): the front stays in HBM/L2 -- panel (m x k, k <= 66) in the L storage, the
// (m-k)^2 contribution block in the cb arena -- and is processed by four launches per tree level:
//   k_big_assemble  one wavefront per front column: zero, scatter A, extend-add children (deterministic)
//   k_big_diag      one workgroup per front: k x k pivot block to LDS, Bunch-Kaufman LDL^T (ldlt_lds)
//   k_big_trsm      64 rows per wavefront: L21 = A21 P L11^{-T} D^{-1}, W21 = L21 D kept for the update
//   k_big_schur     T -= L21 W21^T on 64x64 tiles, v_mfma_f64_16x16x4_f64 (the frontal GEMM)
// ================================================================================================
typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double double2a __attribute__((ext_vector_type(2), aligned(8)));      // two consecutive doubles, 8-byte aligned: one 16-byte access

__global__ __launch_bounds__(256) void k_big_assemble(DevView V, int list_off, int top_mode)
{
    const FrontMeta M = V.fmeta[list_off + blockIdx.y];
    if (M.selfasm) return;                    // pure in-place chain link: its A entries are added by its own pivot-block / TRSM kernels
    const int s = M.s, c0 = M.c0, k = M.k, r0 = M.r0, m = M.m; (void)s; (void)c0; (void)r0; (void)k;
    (void)0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fc = blockIdx.x * 4 + wave;                 // front column owned by this wavefront
    const bool active = fc < m;
    double* col = nullptr;                                 // col[i] = front(i, fc) for i in [0,m) (panel) or [k,m) (T)
    const bool from_arena = top_mode && V.arena && V.arena_off[s] >= 0;   // replicated front at a subtree join
    const bool skip_owned = top_mode && V.arena;                           // rank-owned children are inside the arena
    if (active) {
        const double* Ar = from_arena ? V.arena + V.arena_off[s] + (size_t)fc * m : nullptr;   // all-reduced square, lower part
        if (fc < k) col = V.L + M.panel_off + (size_t)fc * M.ldp;
        else        col = V.cb + M.cb_off + (size_t)(fc - k) * M.ldt - k;
        if (M.alias) {          // the front already sits in its chain child's contribution block: nothing to clear or copy
            if (from_arena) for (int i = fc + lane; i < m; i += 64) col[i] += Ar[i];
        } else if (V.asm_pull) {
            // PULL: every entry of the column is the sum of what the children hold for it -- looked up through the inverse row maps -- and is
            // written ONCE: no zero fill, no read-modify-write chain per child (the scatter form costs three stores and two dependent loads per
            // entry).  Children are added in their fixed order; up to 4 of them here, the others (rare) by the scatter loop below.
            const double* Cc[4]; const int* Iv[4]; int nc = 0;
            for (int cp = M.ch0; cp < M.ch1 && nc < 4; ++cp) {
                const ChildMeta Cm = V.cmeta[cp];
                if (Cm.aliased || (skip_owned && Cm.owner >= 0)) continue;
                const int lo = V.relinv[Cm.inv + fc];
                if (lo < 0) continue;
                Cc[nc] = V.cb + Cm.cb_off + (size_t)lo * Cm.ldt; Iv[nc] = V.relinv + Cm.inv; ++nc;
            }
            if (fc < k) for (int i = lane; i < fc; i += 64) col[i] = 0.0;
            for (int i0 = fc + lane; i0 < m; i0 += 256) {
                int av[4][4]; double t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + 64 * u;
#pragma unroll
                    for (int c = 0; c < 4; ++c) av[c][u] = (c < nc && i < m) ? Iv[c][i] : -1;
                    t[u] = (from_arena && i < m) ? Ar[i] : 0.0;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (av[c][u] >= 0) t[u] += Cc[c][av[c][u]];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int i = i0 + 64 * u; if (i < m) col[i] = t[u]; }
            }
        } else if (fc < k) {
            for (int i = lane; i < m; i += 64) col[i] = (from_arena && i >= fc) ? Ar[i] : 0.0;
        } else {
            for (int i = fc + lane; i < m; i += 64) col[i] = from_arena ? Ar[i] : 0.0;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");      // each wavefront owns its column: ordering of its own stores/loads is all that is needed
    if (active && fc < k) {
        const int q0 = V.acolptr[c0 + fc], q1 = V.acolptr[c0 + fc + 1];
        for (int q = q0 + lane; q < q1; q += 64) col[V.apos[q] - fc * m] += V.aval[q];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");      // each wavefront owns its column: ordering of its own stores/loads is all that is needed
    int npulled = 0;                                           // contributing children the pull loop has dealt with
    for (int cp = M.ch0; cp < M.ch1; ++cp) {
        const ChildMeta Cm = V.cmeta[cp];
        const int ch = Cm.ch; (void)ch;
        if (active && !Cm.aliased && !(skip_owned && Cm.owner >= 0)) {
            const int mc = Cm.mc;
            const int* relc = V.rel + Cm.relbase;
            const int lo = V.relinv[Cm.inv + fc];          // index of parent column fc among the child's update rows (one load, no search)
            if (lo >= 0 && V.asm_pull && !M.alias && npulled < 4) { ++npulled; continue; }
            if (lo >= 0) {
                const double* C = V.cb + Cm.cb_off + (size_t)lo * Cm.ldt;
                int a = lo + lane;
                for (; a + 192 < mc; a += 256) {             // 4 independent gather-add chains in flight per lane
                    const int r0_ = relc[a], r1_ = relc[a + 64], r2_ = relc[a + 128], r3_ = relc[a + 192];
                    const double c0_ = C[a], c1_ = C[a + 64], c2_ = C[a + 128], c3_ = C[a + 192];
                    const double t0 = col[r0_], t1 = col[r1_], t2 = col[r2_], t3 = col[r3_];
                    col[r0_] = t0 + c0_; col[r1_] = t1 + c1_; col[r2_] = t2 + c2_; col[r3_] = t3 + c3_;
                }
                for (; a < mc; a += 64) col[relc[a]] += C[a];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");      // each wavefront owns its column: ordering of its own stores/loads is all that is needed
    }
}

// rows below the pivot block:  W21 = (A21 P) L11^{-T},  then L21 = W21 D^{-1}.  64 rows per workgroup, 16 rows per wavefront,
// BLOCKED SUBSTITUTION with L11 itself, 16 columns at a time: the columns already solved are applied by fp64 MFMA, the 16 x 16
// diagonal block through its inverse (each wavefront inverts one diagonal block first, a 16-lane register substitution), again by
// MFMA -- NOT a product with the whole L11^{-1}: that inverse (needed by the triangular solves only) then leaves the critical
// path of the factorisation, the pivot-block workgroup builds it while the panel is being solved.  In the column loop every
// wavefront works on its own 16 rows: no workgroup barrier between the column blocks.  k <= 128.
struct TrsmLds { double* As; double* Ls; double* Is; double* Ds; int* Ts; int* Lp; double* Au; int kp16, ldl; };
__device__ __forceinline__ TrsmLds trsm_layout(char* smem_raw, const int k, const bool staged, const bool no_ls = false)
{
    TrsmLds T;
    T.kp16 = (k + 15) & ~15; T.ldl = T.kp16 | 1;
    T.As = reinterpret_cast<double*>(smem_raw);               // 64 x kp16: As[r + p*65] = (A21 P)(ibase+r, p), overwritten by W in place
    T.Ls = T.As + (size_t)65 * T.kp16;                        // L11 (strictly lower part, pivot order), zero padded to kp16 x kp16 -- staged only for
    const size_t lsz = no_ls ? 0 : (size_t)T.ldl * T.kp16;            // k <= 64; the 128-column panels of the wide_panels option read it from L2 (LDS budget)
    T.Is = T.Ls + lsz;                     // inverses of the 16 x 16 diagonal blocks of L11: Is[b*272 + i + p*17]
    T.Ds = T.Is + (size_t)17 * T.kp16;                        // dinv[k], doff[k]
    T.Ts = reinterpret_cast<int*>(T.Ds + 2 * k);              // ptype[k]
    T.Lp = T.Ts + k;                                          // lperm[k]
    T.Au = staged ? reinterpret_cast<double*>(T.Lp + k + (k & 1)) : nullptr;   // 64 x k: my rows as they lie in the panel (unpermuted), staged before the pivot block is known
    return T;
}
// (host side: bytes of the layout above)
static size_t trsm_lds_bytes(int k, bool staged)       // (levels with k > 64 launch the variant without the LDS copy of L11)
{
    const size_t kp16 = (size_t)((k + 15) & ~15), ldl = kp16 | 1;
    return (65 * kp16 + (kp16 <= 64 ? ldl * kp16 : 0) + 17 * kp16 + 2 * (size_t)k) * sizeof(double) + (size_t)(2 * k + 2) * sizeof(int) + (staged ? (size_t)65 * kp16 * sizeof(double) : 0) + 16;
}
template <bool STAGED_L>       // L11 staged in LDS (k <= 64) or read from L2 (the 128-column panels of the wide_panels option)
__device__ __forceinline__ void trsm_rows_impl(const DevView& V, const FrontMeta& M, const TrsmLds& T, const int ibase, const int rlim = 64, unsigned long long* ts = nullptr, const bool in_as = false, const int store_mode = 3)
// rlim: rows of the block that are this workgroup's; ts: phase clocks (development); in_as: the caller has staged the rows in As (panel column order);
// store_mode: bit 0 = W21 -> wbuf, bit 1 = L21 -> panel (a caller on the critical chain stores L21 later, out of T.Au: trsm_store_l)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = M.s, c0 = M.c0, k = M.k, m = M.m;
    double* P = V.L + M.panel_off;
    const size_t ldp = (size_t)M.ldp;
    double* W = V.wbuf + M.wb;
    double* As = T.As; const double* Ls = T.Ls; const double* Ds = T.Ds; const int* Ts = T.Ts;
    const int kp16 = T.kp16, ldl = T.ldl;
    auto Lat = [&](int i, int c) -> double {          // L11(i, c), strictly lower part, zero elsewhere
        if (STAGED_L) return Ls[i + c * ldl];
        return (i < k && c < k && i > c) ? P[i + (size_t)c * ldp] : 0.0;
    };
    // pivot data + L11 (written by the pivot-block workgroup / kernel)
    for (int j = tid; j < k; j += 256) { T.Ds[j] = V.dinv[c0 + j]; T.Ds[k + j] = V.doff[c0 + j]; T.Ts[j] = V.ptype[c0 + j]; T.Lp[j] = V.lperm[c0 + j]; }
    // the diagonal-block inverses the blocked pivot-block factorisation left behind (fetched in the same batch; used when valid)
    const int his = V.hasis[s];
    double isv[5];
    {
        const double* Ig = V.isg + (size_t)M.bigidx * ISG_STRIDE;
        const int nis = (kp16 >> 4) * 272;
#pragma unroll
        for (int u = 0; u < 5; ++u) { const int idx = tid + 256 * u; isv[u] = (idx < nis) ? Ig[idx] : 0.0; }
    }
    if (STAGED_L) {          // one batch of independent loads (a dependent global access behind the flag costs ~2 us)
        const int i = tid & 63, cq = tid >> 6;
        double lv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int c = cq + 4 * u; lv[u] = (i < k && c < k && i > c) ? P[i + (size_t)c * ldp] : 0.0; }
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int c = cq + 4 * u; if (i < kp16 && c < kp16) T.Ls[i + c * ldl] = lv[u]; }
    }      // (k > 64: read through Lat() from L2)
    if (his) {
        const int nis = (kp16 >> 4) * 272;
#pragma unroll
        for (int u = 0; u < 5; ++u) { const int idx = tid + 256 * u; if (idx < nis) T.Is[idx] = isv[u]; }
    }
    __syncthreads();
    if (ts) ts[0] = clock64();
#ifdef MI355X_PIVSTAT
    const bool bprobe = gridDim.x == 1 && blockIdx.y == 1 && tid == 0 && T.Au;
    if (bprobe) g_dt[10] = wall_clock64();
#endif
    if (in_as) {
        // the rows sit in As already, in PANEL column order: nothing to do behind the blocked factorisation (natural pivot order), a
        // column permutation in place behind the strict loop
        if (!his) {
            const int r = tid & 63, pq = tid >> 6;
            double av[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int p = pq + 4 * u; av[u] = (p < k) ? As[r + T.Lp[p] * 65] : 0.0; }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int p = pq + 4 * u; if (p < kp16) As[r + p * 65] = av[u]; }
        }
    }
    else if (T.Au && kp16 <= 64) {      // (batched: the three LDS accesses of an element are a dependent chain)
        const int r = tid & 63, pq = tid >> 6;
        int lpv[16]; double av[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int p = pq + 4 * u; lpv[u] = (p < k) ? T.Lp[p] : -1; }
#pragma unroll
        for (int u = 0; u < 16; ++u) av[u] = (lpv[u] >= 0) ? T.Au[r + lpv[u] * 65] : 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int p = pq + 4 * u; if (p < kp16) As[r + p * 65] = av[u]; }
    }
    else if (T.Au) { for (int idx = tid; idx < 64 * kp16; idx += 256) { const int r = idx & 63, p = idx >> 6; As[r + p * 65] = (p < k) ? T.Au[r + T.Lp[p] * 65] : 0.0; } }
    else      { for (int idx = tid; idx < 64 * kp16; idx += 256) { const int r = idx & 63, p = idx >> 6; As[r + p * 65] = (p < k && ibase + r < m) ? P[ibase + r + (size_t)T.Lp[p] * ldp] : 0.0; } }
    // inverses of the unit-lower 16 x 16 diagonal blocks: block b by wavefront b & 3, column c of the inverse by lane c
    // (only after the strict pivot loop: the blocked factorisation hands them over)
    for (int b = wave; 16 * b < kp16 && !his; b += 4) {
        if (lane < 16) {
            const int o = 16 * b;
            double x[16];                    // x = column `lane` of the inverse; column-oriented substitution: independent updates per step
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = (i == lane) ? 1.0 : 0.0;
#pragma unroll
            for (int pp = 0; pp < 15; ++pp) {
                const double xp = (pp >= lane) ? x[pp] : 0.0;
#pragma unroll
                for (int i = pp + 1; i < 16; ++i) x[i] = fma(-Lat(o + i, o + pp), xp, x[i]);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) T.Is[b * 272 + i + lane * 17] = x[i];
        }
    }
    __syncthreads();
    if (ts) ts[1] = clock64();
#ifdef MI355X_PIVSTAT
    if (bprobe) g_dt[11] = wall_clock64();
#endif
    const int l15 = lane & 15, l4 = lane >> 4;
    const int r16 = wave * 16;
    for (int c16 = 0; c16 < kp16; c16 += 16) {
        // columns [c16, c16+16) minus what the solved columns contribute: (16 x c16) . (c16 x 16), transposed product as in the updates
        v4f64 acc = (v4f64){0.0, 0.0, 0.0, 0.0};
        if (c16 <= 48) {     // operands of the whole product in flight at once
            double oa[12], ob[12];
#pragma unroll
            for (int u = 0; u < 12; ++u) { const bool v = 4 * u < c16; oa[u] = v ? Lat(c16 + l15, 4 * u + l4) : 0.0; ob[u] = v ? As[r16 + l15 + (4 * u + l4) * 65] : 0.0; }
#pragma unroll
            for (int u = 0; u < 12; ++u) if (4 * u < c16) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(oa[u], ob[u], acc, 0, 0, 0);
        } else {
            for (int p = 0; p < c16; p += 4)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Lat(c16 + l15, p + l4), As[r16 + l15 + (p + l4) * 65], acc, 0, 0, 0);
        }
        // the block A' = A - acc goes from accumulator layout to operand layout through LDS (rows of this wavefront only)
#pragma unroll
        for (int g = 0; g < 4; ++g) As[r16 + l15 + (c16 + l4 + 4 * g) * 65] -= acc[g];
        double bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) bv[u] = As[r16 + l15 + (c16 + 4 * u + l4) * 65];
        const double* Ib = T.Is + (c16 >> 4) * 272;
        v4f64 w = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 4; ++u) w = __builtin_amdgcn_mfma_f64_16x16x4f64(Ib[l15 + (4 * u + l4) * 17], bv[u], w, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) As[r16 + l15 + (c16 + l4 + 4 * g) * 65] = w[g];
    }
    __syncthreads();
    if (ts) ts[2] = clock64();
#ifdef MI355X_PIVSTAT
    if (bprobe) g_dt[12] = wall_clock64();
#endif
    // a posteriori threshold test on the rows below the pivot block (the in-block test of ldlt_reg cannot see them): a column
    // with a multiplier above 1/u is a FAILED pivot -- a delayed pivot in MA97/SSIDS, counted once per column here (num_delay)
    // (two consecutive rows per lane: 16-byte stores -- a CU issues stores at ~10 bytes per cycle whatever their width.  The loop-invariant
    //  pieces of the view are pinned in VGPRs: with ~100 SGPRs of kernel arguments spilled, the compiler otherwise RELOADS them from the
    //  kernarg segment inside the loop, a scalar-cache round trip per iteration)
    {
        const int r = 2 * (tid & 31), jq = tid >> 5;
        const int i = ibase + r;
        const bool ok0 = i < m && r < rlim, ok1 = i + 1 < m && r + 1 < rlim;
        double* Wv = W + i; double* Pv = P + i; double uv = V.pivtol; double* Auv = T.Au ? T.Au + r : nullptr; const double* Asv = As + r;
        asm volatile("" : "+v"(Wv), "+v"(Pv), "+v"(uv), "+v"(Auv), "+v"(Asv));
        if (Auv) for (int j = k + jq; j < kp16; j += 8) { Auv[j * 65] = 0.0; Auv[1 + j * 65] = 0.0; }
        bool big = false;
        if (his) {        // behind the blocked factorisation every pivot is 1x1: straight-line body
#pragma unroll 4
            for (int j = jq; j < k; j += 8) {
                const double d = Ds[j];
                const double w0 = Asv[j * 65], w1 = Asv[1 + j * 65];
                const double l0 = w0 * d, l1 = w1 * d;
                if (Auv) { Auv[j * 65] = l0; Auv[1 + j * 65] = l1; }
                if (ok1) {
                    if (store_mode & 1) *reinterpret_cast<double2a*>(&Wv[(size_t)j * m]) = (double2a){w0, w1};
                    if (store_mode & 2) *reinterpret_cast<double2a*>(&Pv[(size_t)j * ldp]) = (double2a){l0, l1};
                } else if (ok0) {
                    if (store_mode & 1) Wv[(size_t)j * m] = w0;
                    if (store_mode & 2) Pv[(size_t)j * ldp] = l0;
                }
                big |= (ok0 && fabs(l0) * uv > 1.0) || (ok1 && fabs(l1) * uv > 1.0);
            }
        } else {
            for (int j = jq; j < k; j += 8) {
                const int pt = Ts[j];
                const double w0 = Asv[j * 65], w1 = Asv[1 + j * 65];
                double l0, l1;
                if (pt == 1) { l0 = w0 * Ds[j]; l1 = w1 * Ds[j]; }
                else if (pt == 2) { l0 = Ds[j] * w0 + Ds[k + j] * Asv[(j + 1) * 65]; l1 = Ds[j] * w1 + Ds[k + j] * Asv[1 + (j + 1) * 65]; }
                else { l0 = Ds[k + j - 1] * Asv[(j - 1) * 65] + Ds[j] * w0; l1 = Ds[k + j - 1] * Asv[1 + (j - 1) * 65] + Ds[j] * w1; }
                if (Auv) { Auv[j * 65] = l0; Auv[1 + j * 65] = l1; }      // (the staging copy is dead: L21 of my rows stays in LDS for the updates that follow)
                if (ok1) {
                    if (store_mode & 1) *reinterpret_cast<double2a*>(&Wv[(size_t)j * m]) = (double2a){w0, w1};
                    if (store_mode & 2) *reinterpret_cast<double2a*>(&Pv[(size_t)j * ldp]) = (double2a){l0, l1};
                } else if (ok0) {
                    if (store_mode & 1) Wv[(size_t)j * m] = w0;
                    if (store_mode & 2) Pv[(size_t)j * ldp] = l0;
                }
                big |= (ok0 && fabs(l0) * uv > 1.0) || (ok1 && fabs(l1) * uv > 1.0);
            }
        }
        // a multiplier above 1/u somewhere in my rows (rare): find the columns and count each once
        if (__ballot(big) != 0ull) {
            for (int j = jq; j < k; j += 8) {
                const int pt = Ts[j];
                double l0, l1;
                const double w0 = Asv[j * 65], w1 = Asv[1 + j * 65];
                if (pt == 1) { l0 = w0 * Ds[j]; l1 = w1 * Ds[j]; }
                else if (pt == 2) { l0 = Ds[j] * w0 + Ds[k + j] * Asv[(j + 1) * 65]; l1 = Ds[j] * w1 + Ds[k + j] * Asv[1 + (j + 1) * 65]; }
                else { l0 = Ds[k + j - 1] * Asv[(j - 1) * 65] + Ds[j] * w0; l1 = Ds[k + j - 1] * Asv[1 + (j - 1) * 65] + Ds[j] * w1; }
                if (((ok0 && fabs(l0) * uv > 1.0) || (ok1 && fabs(l1) * uv > 1.0)) && atomicExch(&V.colfail[c0 + j], 1) == 0) atomicAdd(&V.fstat[s].w, 1);
            }
        }
    }
    if (ts) ts[3] = clock64();
}
// L21 of a row block out of its LDS copy (T.Au) into the panel -- for a caller that kept it back (store_mode 1)
__device__ __forceinline__ void trsm_store_l(const DevView& V, const FrontMeta& M, const double* Lr, const int ibase, const int rlim)
{
    const int tid = threadIdx.x, k = M.k, m = M.m;
    double* P = V.L + M.panel_off;
    const size_t ldp = (size_t)M.ldp;
    const int r = 2 * (tid & 31), jq = tid >> 5, i = ibase + r;
    const bool ok0 = i < m && r < rlim, ok1 = i + 1 < m && r + 1 < rlim;
    for (int j = jq; j < k; j += 8) {
        if (ok1) *reinterpret_cast<double2a*>(&P[i + (size_t)j * ldp]) = (double2a){Lr[r + j * 65], Lr[r + 1 + j * 65]};
        else if (ok0) P[i + (size_t)j * ldp] = Lr[r + j * 65];
    }
}
template <bool WIDEK>          // WIDEK: the level has panels of more than 64 columns (wide_panels option): L11 is not staged in LDS
__global__ __launch_bounds__(256) void k_big_trsm(DevView V, int list_off, int rb0)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x;
    const FrontMeta M = V.fmeta[list_off + blockIdx.y];
    const int k = M.k, m = M.m;
    const int ibase = k + ((int)blockIdx.x + rb0) * 64;       // rb0: first row block of this launch (chain look-ahead: block 0 alone, then the rest)
    if (ibase >= m) return;
    double* P = V.L + M.panel_off;
    const size_t ldp = (size_t)M.ldp;
    if (M.selfasm) {            // pure in-place chain link: the A entries of the rows below are added by the workgroup that owns the rows
        for (int q = M.aq0 + tid; q < M.aq1; q += 256) { const int pos = V.apos[q]; const int i = pos % m, c = pos / m; if (i >= ibase && i < ibase + 64) P[i + (size_t)c * ldp] += V.aval[q]; }
        __syncthreads();
    }
    const TrsmLds T = trsm_layout(smem_raw, k, false, WIDEK);
    trsm_rows_impl<!WIDEK>(V, M, T, ibase);
}

// 64 x 64 tile of the trailing update on one 256-thread workgroup (k_big_schur64; also the narrow updates fused into k_big_diag_trsm)
__device__ __forceinline__ void schur64_tile(const DevView& V, const FrontMeta& M, const int t, const int mode)      // mode 0: every tile, 1: tile (0,0) only (look-ahead), 2: all but it
{
    const int k = M.k, m = M.m;
    const int mu = m - k;
    const int nt = (mu + 63) >> 6;
    if ((mode == 2 && t == 0) || (mode == 1 && t != 0)) return;
    int ti, tc, climit, j0;
    if (M.grem > 0) {
        const int ntc = (M.grem + 63) >> 6;
        if (t >= nt * ntc) return;
        ti = t / ntc; tc = t - ti * ntc; climit = M.grem; j0 = M.gpos;
    } else {
        if (t >= nt * (nt + 1) / 2) return;
        ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while (ti * (ti + 1) / 2 > t) --ti;
        while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
        tc = t - ti * (ti + 1) / 2; climit = mu; j0 = 0;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = ti * 64 + (wave >> 1) * 32, cc0 = tc * 64 + (wave & 1) * 32;
    if (i0 + 31 < cc0 || cc0 >= climit || i0 >= mu) return;
    const int l15 = lane & 15, l4 = lane >> 4;
    v4f64 acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q) acc[r][q] = (v4f64){0.0, 0.0, 0.0, 0.0};
    const int ca = cc0 + l15, cb_ = cc0 + 16 + l15, ia = i0 + l15, ib = i0 + 16 + l15;
    for (int j = j0; j <= M.gpos; ++j) {
        const GroupLink G = V.gtab[M.gbase + j];
        const int kj = G.k;
        const double* Lp = V.L + G.panel_off + (G.m - mu);
        const double* Wp = V.wbuf + G.wb + (G.m - mu);
        for (int p = 0; p < kj; p += 4) {
            const int pk = p + l4;
            const bool v = pk < kj;
            const size_t off = (size_t)pk * G.m, offp = (size_t)pk * G.ldp;
            const double a0 = (v && ca < mu) ? Wp[ca + off] : 0.0;
            const double a1 = (v && cb_ < mu) ? Wp[cb_ + off] : 0.0;
            const double b0 = (v && ia < mu) ? Lp[ia + offp] : 0.0;
            const double b1 = (v && ib < mu) ? Lp[ib + offp] : 0.0;
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    double* T = V.cb + M.cb_off;
    double tv[2][2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = cc0 + r * 16 + l4 + 4 * g, i = i0 + q * 16 + l15;
                tv[r][q][g] = (i < mu && c < climit && i >= c) ? T[i + (size_t)c * M.ldt] : 0.0;
            }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = cc0 + r * 16 + l4 + 4 * g, i = i0 + q * 16 + l15;
                if (i < mu && c < climit && i >= c) T[i + (size_t)c * M.ldt] = tv[r][q][g] - acc[r][q][g];
            }
}


// the same tile for a NARROW update on the serial chain (this link's k <= 64 columns only): every operand of the product is
// requested before the first MFMA -- behind the panel workgroups' flag each dependent access is a ~2 us trip to another XCD's data
__device__ __forceinline__ void narrow_tile64(const DevView& V, const FrontMeta& M, const int t)
{
    const int k = M.k, m = M.m;
    const int mu = m - k;
    const int nt = (mu + 63) >> 6, ntc = (M.grem + 63) >> 6;
    if (t >= nt * ntc) return;
    const int ti = t / ntc, tc = t - ti * ntc, climit = M.grem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = ti * 64 + (wave >> 1) * 32, cc0 = tc * 64 + (wave & 1) * 32;
    if (i0 + 31 < cc0 || cc0 >= climit || i0 >= mu) return;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int ca = cc0 + l15, cb_ = cc0 + 16 + l15, ia = i0 + l15, ib = i0 + 16 + l15;
    const GroupLink G = V.gtab[M.gbase + M.gpos];
    const int kj = G.k;
    const double* Lp = V.L + G.panel_off + (G.m - mu);
    const double* Wp = V.wbuf + G.wb + (G.m - mu);
    double a0[16], a1[16], b0[16], b1[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int pk = 4 * u + l4;
        const bool v = pk < kj;
        const size_t off = (size_t)pk * G.m, offp = (size_t)pk * G.ldp;
        a0[u] = (v && ca < mu) ? Wp[ca + off] : 0.0;
        a1[u] = (v && cb_ < mu) ? Wp[cb_ + off] : 0.0;
        b0[u] = (v && ia < mu) ? Lp[ia + offp] : 0.0;
        b1[u] = (v && ib < mu) ? Lp[ib + offp] : 0.0;
    }
    double* T = V.cb + M.cb_off;
    double tv[2][2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = cc0 + r * 16 + l4 + 4 * g, i = i0 + q * 16 + l15;
                tv[r][q][g] = (i < mu && c < climit && i >= c) ? T[i + (size_t)c * M.ldt] : 0.0;
            }
    v4f64 acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q) acc[r][q] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u < 16; ++u)
        if (4 * u < kj) {
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b0[u], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b1[u], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b0[u], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b1[u], acc[1][1], 0, 0, 0);
        }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = cc0 + r * 16 + l4 + 4 * g, i = i0 + q * 16 + l15;
                if (i < mu && c < climit && i >= c) T[i + (size_t)c * M.ldt] = tv[r][q][g] - acc[r][q][g];
            }
}

// Pivot block AND panel solve of a front in ONE launch (the top of the tree, where a level has a handful of fronts and both
// kernels are a chain of dependent round trips rather than work): workgroup 0 of a front is the pivot-block kernel and raises the
// front's flag as soon as L11 and D are stored (its inverse follows, off the critical path); workgroups 1.. each own 64 panel
// rows, add their A entries and stage their rows in LDS BEFORE they wait, then permute / solve / scale.  Same arithmetic as
// k_big_diag_reg + k_big_trsm, bit for bit.
__global__ __launch_bounds__(256) void k_big_diag_trsm(DevView V, int list_off, int nrb)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const FrontMeta M = V.fmeta[list_off + blockIdx.x];      // x = front, y = role: every pivot block is dispatched before the first waiting workgroup
    const int role = blockIdx.y;
    const int epoch = V.sepoch[2];
#ifdef MI355X_PIVSTAT
    const bool tprobe = gridDim.x == 1 && threadIdx.x == 0;
    if (tprobe && role == 0) g_dt[0] = wall_clock64();
    if (tprobe && role == 1) g_dt[6] = wall_clock64();
#endif
    if (role == 0) { big_diag_body<4, 256>(V, M, smem_raw, &V.sflag_d[M.s], epoch); return; }
    const int tid = threadIdx.x;
    const int s = M.s, k = M.k, m = M.m;
    if (role > nrb) {
        // workgroups behind the panel blocks: the NARROW trailing update of a chain link that is not the last of its group (64 x 64
        // tiles over the group's remaining panel columns: a fraction of a GFlop, not worth a launch of its own on the serial chain)
        const int t = role - 1 - nrb;
        const int mu = m - k, nt = (mu + 63) >> 6;
        if (M.grem <= 0 || t >= nt * ((M.grem + 63) >> 6)) return;
        if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(&V.tcnt[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nt) { __builtin_amdgcn_s_sleep(16);      // (~0.4 us: a hundred pollers of one address must not saturate its memory channel)
                if (++spins > (1 << 24)) { V.qstat[1] = 1; break; } }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        narrow_tile64(V, M, t);
        return;
    }
    const int ibase = k + (role - 1) * 64;
    if (ibase >= m) return;
    double* P = V.L + M.panel_off;
    const size_t ldp = (size_t)M.ldp;
    if (M.selfasm) {
        for (int q = M.aq0 + tid; q < M.aq1; q += 256) { const int pos = V.apos[q]; const int i = pos % m, c = pos / m; if (i >= ibase && i < ibase + 64) P[i + (size_t)c * ldp] += V.aval[q]; }
        __syncthreads();
    }
    const TrsmLds T = trsm_layout(smem_raw, k, true);
    for (int idx = tid; idx < 64 * k; idx += 256) { const int r = idx & 63, c = idx >> 6; T.Au[r + c * 65] = (ibase + r < m) ? P[ibase + r + (size_t)c * ldp] : 0.0; }
#ifdef MI355X_PIVSTAT
    if (tprobe && role == 1) g_dt[7] = wall_clock64();
#endif
    chain_wait(&V.sflag_d[s], epoch, V.qstat + 1);            // L11 and D of this front are stored
#ifdef MI355X_PIVSTAT
    if (tprobe && role == 1) g_dt[8] = wall_clock64();
#endif
    trsm_rows_impl<true>(V, M, T, ibase);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // my rows of L21 / W21 are stored: one more panel block done
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(&V.tcnt[s], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef MI355X_PIVSTAT
    if (tprobe && role == 1) {
        const unsigned long long te = wall_clock64(), t0 = g_dt[0];
        for (int q = 1; q <= 8; ++q) if (q != 3 && q != 4) g_dtacc[q] += g_dt[q] - t0;
        g_dtacc[9] += te - t0; g_dtacc[0] += 1;
        for (int q = 10; q <= 12; ++q) g_dtacc[q] += g_dt[q] - t0;
    }
#endif
}

// T(i,c) -= sum_p L21(i,p) W21(c,p),  i >= c, on 64x64 tiles; each of the 4 waves owns a 32x32 sub-tile made of
// 2x2 v_mfma_f64_16x16x4_f64 accumulators.  The product is formed TRANSPOSED (A operand = W rows, B operand = L rows)
// so that the 16 lanes sharing an accumulator register hold 16 consecutive ROWS of the column-major T => 128-byte
// coalesced read-modify-write segments.
// Chain groups: a link that is not the last of its group only updates the group's remaining `grem` columns (the panels
// of the later links); the LAST link applies the update of ALL the group's panels to its contribution block in one
// pass (K = sum of the links' columns, <= 256), so the block is read and written once per group instead of once per link.
// Look-ahead: the update of a group-last front whose chain continues may be SPLIT (FrontMeta::split): part 1 = the first two
// tile columns (all that the next group's panels, pivot blocks and narrow updates touch) stays on the main stream, part 2 =
// the rest runs on a second stream, overlapped with the latency-bound pivot chains of the next group.  part 0 = everything.
constexpr int SCHUR_KC = 16, SCHUR_LD = 132;
__device__ __forceinline__ void schur_tile(const DevView& V, const FrontMeta& M, const int t, const int part, const int skip00,
                                           double (&As)[2][SCHUR_KC][SCHUR_LD], double (&Bs)[2][SCHUR_KC][SCHUR_LD])
{
    // 128 x 128 tile per workgroup of 16 wavefronts (32 x 32 each = 2 x 2 accumulators of v_mfma_f64_16x16x4_f64).  The
    // operands are staged through LDS in chunks of 16 panel columns (double buffered, one barrier per chunk): every panel
    // entry is fetched from L2 once per tile instead of once per wavefront, and 4 wavefronts per SIMD hide the LDS latency.
    constexpr int KC = SCHUR_KC;
    const int k = M.k, m = M.m;
    const int mu = m - k;
    const int nt = (mu + 127) >> 7;
    int ti, tc, climit, j0;
    const bool sp = part != 0 && M.split != 0;
    if (part == 2 && !sp) return;
    if (M.grem > 0) {
        const int ntc = (M.grem + 127) >> 7;
        if (t >= nt * ntc) return;
        ti = t / ntc; tc = t - ti * ntc; climit = M.grem; j0 = M.gpos;
        if (ti < tc) return;
    } else if (sp && part == 1) {
        if (t >= 2 * nt - 1) return;
        if (t < nt) { ti = t; tc = 0; } else { ti = t - nt + 1; tc = 1; }
        climit = mu; j0 = 0;
    } else {
        const int n2 = sp ? nt - 2 : nt, sh = sp ? 2 : 0;
        const int ntri = n2 * (n2 + 1) / 2;
        const int tab = sp ? M.ttab2 : M.ttab;
        if (tab >= 0) {
            // large update: workgroups are dealt to the 8 XCDs round-robin (linear id mod 8), so XCD x walks the x-th
            // contiguous eighth of a super-tile ordered list (8 x 8 tile blocks): the ~64 tiles in flight on one XCD share
            // 16 panel row blocks, which its 4 MB L2 holds, instead of streaming the whole panel from the fabric per tile
            const int chunk = (ntri + 7) >> 3;
            const int pos = (t & 7) * chunk + (t >> 3);
            if ((t >> 3) >= chunk || pos >= ntri) return;
            const int e = V.tile_tab[tab + pos];
            ti = e >> 16; tc = e & 0xffff;
        } else {
            if (t >= ntri) return;
            ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
            while (ti * (ti + 1) / 2 > t) --ti;
            while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
            tc = t - ti * (ti + 1) / 2;
        }
        climit = mu; j0 = 0;
        ti += sh; tc += sh;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int wr = (wave >> 2) * 32, wc = (wave & 3) * 32;             // this wavefront's 32 x 32 block inside the tile
    const int i0 = ti * 128, cc0 = tc * 128;
    // skip00: the leading 64 x 64 block of the trailing matrix (the NEXT link's pivot block) was already updated by the look-ahead launch
    const bool work = (i0 + wr + 31 >= cc0 + wc) && (cc0 + wc < climit) && (i0 + wr < mu) && !(skip00 && i0 + wr < 64 && cc0 + wc < 64);
    v4f64 acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q) acc[r][q] = (v4f64){0.0, 0.0, 0.0, 0.0};
    // staging role of this thread: row (tid & 127) of the tile, panel columns (tid >> 7) and (tid >> 7) + 8 of the chunk
    const int srow = tid & 127, scol = tid >> 7;
    const bool arow_ok = cc0 + srow < mu, brow_ok = i0 + srow < mu;
    // software pipeline of depth 2 over the 16-column chunks: while chunk i is multiplied out of LDS, chunk i+1 sits in registers on its
    // way to LDS and chunk i+2 is in flight from L2 / HBM -- on the chain levels a launch is a handful of workgroups and the per-chunk
    // cost is the load latency, not the 16 MFMAs; the link record (gtab) is re-read only when the link changes, not once per chunk.
    // (The order in which a tile element accumulates its products is unchanged: results are bit-identical.)
    GroupLink Gc = V.gtab[M.gbase + j0];
    int gj = j0;
    double ga0[2], gb0[2], ga1[2], gb1[2];                 // two register stages (statically named: no indexed register arrays)
    auto fetch = [&](int j, int p0, double (&ga)[2], double (&gb)[2]) {
        if (j != gj) { Gc = V.gtab[M.gbase + j]; gj = j; }
        const double* Lp = V.L + Gc.panel_off + (Gc.m - mu) + i0 + srow;
        const double* Wp = V.wbuf + Gc.wb + (Gc.m - mu) + cc0 + srow;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pk = p0 + scol + 8 * u;
            const bool v = pk < Gc.k;
            ga[u] = (v && arow_ok) ? Wp[(size_t)pk * Gc.m] : 0.0;
            gb[u] = (v && brow_ok) ? Lp[(size_t)pk * Gc.ldp] : 0.0;
        }
    };
    // chunk sequence: (link j, first column p), j = j0 .. M.gpos, p = 0, KC, ... < k_j.  advance() is called right after the fetch of
    // chunk (j, p), so the cached link record is that of j.
    auto advance = [&](int& j, int& p) -> bool { p += KC; if (p >= Gc.k) { ++j; p = 0; } return j <= M.gpos; };
    auto to_lds = [&](int b, const double (&ga)[2], const double (&gb)[2]) { As[b][scol][srow] = ga[0]; As[b][scol + 8][srow] = ga[1]; Bs[b][scol][srow] = gb[0]; Bs[b][scol + 8][srow] = gb[1]; };
    auto multiply = [&](int b) {
        if (!work) return;
#pragma unroll
        for (int st = 0; st < KC / 4; ++st) {
            const int kk = 4 * st + l4;
            const double a0 = As[b][kk][wc + l15], a1 = As[b][kk][wc + 16 + l15];
            const double b0 = Bs[b][kk][wr + l15], b1 = Bs[b][kk][wr + 16 + l15];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    };
    int jn = j0, pn = 0;                   // the chunk most recently fetched
    fetch(jn, pn, ga0, gb0);
    to_lds(0, ga0, gb0);                   // chunk 0 -> LDS
    bool have1 = advance(jn, pn);          // chunk 1 exists -> stage 0
    if (have1) fetch(jn, pn, ga0, gb0);
    bool have2 = have1 && advance(jn, pn); // chunk 2 exists -> stage 1
    if (have2) fetch(jn, pn, ga1, gb1);
    __syncthreads();
    while (true) {
        multiply(0);
        if (!have1) break;
        to_lds(1, ga0, gb0);
        have1 = have2;
        if (have2) { have2 = advance(jn, pn); if (have2) fetch(jn, pn, ga0, gb0); }
        __syncthreads();
        multiply(1);
        if (!have1) break;
        to_lds(0, ga1, gb1);
        have1 = have2;
        if (have2) { have2 = advance(jn, pn); if (have2) fetch(jn, pn, ga1, gb1); }
        __syncthreads();
    }
    if (!work) return;
    double* T = V.cb + M.cb_off;
    double tv[2][2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = cc0 + wc + r * 16 + l4 + 4 * g;      // D row index  -> T column
                const int i = i0 + wr + q * 16 + l15;              // D column index -> T row
                tv[r][q][g] = (i < mu && c < climit && i >= c) ? T[i + (size_t)c * M.ldt] : 0.0;
            }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = cc0 + wc + r * 16 + l4 + 4 * g;
                const int i = i0 + wr + q * 16 + l15;
                if (i < mu && c < climit && i >= c) T[i + (size_t)c * M.ldt] = tv[r][q][g] - acc[r][q][g];
            }
}
// part 0 / 1: one tile per workgroup.  part 2 (look-ahead, second stream) strides over the tiles, normally also one per
// workgroup (a persistent grid smaller than the chip was measured: one 16-wave workgroup per CU reaches half the MFMA rate).
__global__ __launch_bounds__(1024) void k_big_schur(DevView V, int list_off, int part, int ntiles, int skip00)
{
    __shared__ double As[2][SCHUR_KC][SCHUR_LD];      // W rows (-> T columns) of the tile
    __shared__ double Bs[2][SCHUR_KC][SCHUR_LD];      // L rows (-> T rows)
    const FrontMeta M = V.fmeta[list_off + blockIdx.y];
    if (part != 2) { schur_tile(V, M, blockIdx.x, part, skip00, As, Bs); return; }
    if (!M.split) return;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) { schur_tile(V, M, t, part, 0, As, Bs); __syncthreads(); }
}


// Small-front variant of the trailing update (levels whose largest front has <= 640 rows: thousands of fronts, a handful
// of tiles each): 64 x 64 tile per 256-thread workgroup, 32 x 32 per wavefront, operands straight from L2.  The 128 x 128
// LDS-staged kernel above leaves most of its 16 wavefronts idle on such fronts (PMC: 7.6 % MFMA utilisation).
__global__ __launch_bounds__(256) void k_big_schur64(DevView V, int list_off, int mode)
{
    const FrontMeta M = V.fmeta[list_off + blockIdx.y];
    schur64_tile(V, M, blockIdx.x, mode);
}


// ================================================================================================
// CHAIN GROUPS FACTORED AS ONE UNIT.  A chain group = up to 4 consecutive links of an in-place separator chain (<= 256 columns); every
// link after the first has the chain child as its ONLY child, so the whole group can be factored at the tree level of its first link:
// one launch for the group's leading block and all its panel rows (k_grp_fused), then the rank-(<= 256) update of the contribution block
// by all the group's panels (k_big_schur, unchanged).  (A two-launch variant -- one 512-thread workgroup walking the leading block link by
// link, row blocks behind it -- was built first and measured slower: 210 us per group, a lone wavefront issues one fp64 MFMA per ~150
// cycles; removed.)
// ================================================================================================
// A chain group as ONE launch of row-block workgroups (the default of the grouped schedule): grid = (groups, 4 + row blocks below the
// group).  Role q < 4 owns the pivot rows of the group's link q, role 4 + b the b-th block of 64 rows below the group's columns.  A
// row block walks the links p before its own (all of them for the rows below): rows staged in LDS, wait for link p's pivot block
// (flag), solve against it (k_big_trsm's arithmetic), publish W / L of its rows (flag), then update its own rows' entries in the
// columns of the later links r, waiting per r for the rows that hold W(r, p).  A pivot-row block then factors its own pivot block
// (big_diag_body: blocked a-posteriori LDL^T or the strict loop) and raises the flag the blocks after it wait for.  Every wait is on
// a workgroup with a SMALLER role of the same group: with workgroups dispatched in linear order (roles are the slow grid dimension)
// a waiting workgroup never holds up the one it waits for; the spins are bounded all the same (qstat[1]).
// Critical path per link: pivot block -> flag -> one 64 x 64 panel solve -> one 64 x 64 x 64 update -> next pivot block.
constexpr size_t GRP_DB_BYTES = (size_t)64 * 65 * sizeof(double);      // a pivot-row block's own pivot block, LDS-resident from the start of the launch
__global__ __launch_bounds__(256) void k_grp_fused(DevView V, int list_off, int staged)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const FrontMeta M = V.fmeta[list_off + blockIdx.x];              // the LAST link of the group
    const int g = M.gpos + 1, role = blockIdx.y;
    const int tail = M.m - M.k;                                      // rows below the group's columns
    int myq = -1, e0 = 0;
    if (role < 4) { if (role >= g) return; myq = role; }
    else { e0 = 64 * (role - 4); if (e0 >= tail) return; }
    const int epoch = V.sepoch[2];
#define GSTAMP(i) do { if (V.dbg && blockIdx.x == 0 && role < 4 && tid == 0) V.dbg[32 + 8 * role + (i)] = wall_clock64(); } while (0)
    GSTAMP(0);
    unsigned long long tsv[28];
#pragma unroll
    for (int q = 0; q < 28; ++q) tsv[q] = 0ull;
    unsigned long long* const tsp = (V.dbg && blockIdx.x == 0 && role == 1) ? tsv : nullptr;
    int kk[4], koff[5];
    koff[0] = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { kk[j] = (j < g) ? V.gtab[M.gbase + j].k : 0; koff[j + 1] = koff[j] + kk[j]; }
    const int last_p = (myq >= 0) ? myq - 1 : g - 1;
    int myrows = min(64, tail - e0), mystart = M.gcols + e0;          // my rows, counted from the first column of the group
#pragma unroll
    for (int j = 0; j < 4; ++j) if (j == myq) { myrows = kk[j]; mystart = koff[j]; }
    // ---- a pivot-row block keeps its own pivot block in LDS (lower triangle) from the start: the updates of the links before it
    //      are applied there, the factorisation at the end reads it there -- the block never makes a round trip through L2 ----
    double* Db = reinterpret_cast<double*>(smem_raw);
    const int ldb = myrows | 1;
    GroupLink Gq = V.gtab[M.gbase];
    if (myq >= 0) {
#pragma unroll
        for (int j = 1; j < 4; ++j) if (j == myq) Gq = V.gtab[M.gbase + j];
        const double* Pq = V.L + Gq.panel_off;
        const size_t ldq = (size_t)Gq.ldp;
        const int i = tid & 63, cq = tid >> 6, kq = Gq.k;
        double pv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) { const int c = cq + 4 * e; pv[e] = (i < kq && c < kq && i >= c) ? Pq[i + (size_t)c * ldq] : 0.0; }
#pragma unroll
        for (int e = 0; e < 16; ++e) { const int c = cq + 4 * e; if (i < kq && c < kq && i >= c) Db[i + c * ldb] = pv[e]; }
        __syncthreads();
        if (Gq.selfasm) {
            for (int q = Gq.aq0 + tid; q < Gq.aq1; q += 256) { const int pos = V.apos[q]; const int ii = pos % Gq.m, cc = pos / Gq.m; if (ii < kq) Db[ii + cc * ldb] += V.aval[q]; }
            __syncthreads();
        }
    }
    char* tsm = smem_raw + GRP_DB_BYTES;
    const double* crit_Lr = nullptr; FrontMeta crit_Mp = M; int crit_ibase = 0;      // the kept-back L21 of the last link before my pivot block
    for (int p = 0; p <= last_p; ++p) {
        const GroupLink G = V.gtab[M.gbase + p];
        const int k = G.k, m = G.m;
        int kend = 0;                                                 // columns of the group up to and including link p
#pragma unroll
        for (int j = 0; j < 4; ++j) if (j == p) kend = koff[j + 1];
        const int ibase = k + (mystart - kend);                       // my first row in this link's front
        double* P = V.L + G.panel_off;
        const size_t ldp = (size_t)G.ldp;
        FrontMeta Mp = M;
        Mp.s = G.s; Mp.c0 = G.c0; Mp.k = k; Mp.m = m; Mp.panel_off = G.panel_off; Mp.ldp = G.ldp; Mp.wb = G.wb;
        const TrsmLds T = trsm_layout(tsm, k, staged != 0);           // (levels with hundreds of groups: no staging copy, more workgroups per CU)
        Mp.bigidx = G.bigidx;
        if (staged) {
            // my rows as they lie in the panel (and the link's own A entries for them) go to LDS before the pivot block is known --
            // straight into the working block: behind the blocked factorisation the pivot order is the panel's column order
            for (int idx = tid; idx < 64 * T.kp16; idx += 256) { const int r = idx & 63, c = idx >> 6; T.As[r + c * 65] = (c < k && r < myrows && ibase + r < m) ? P[ibase + r + (size_t)c * ldp] : 0.0; }
            if (G.selfasm) {
                __syncthreads();
                for (int q = G.aq0 + tid; q < G.aq1; q += 256) { const int pos = V.apos[q]; const int i = pos % m, c = pos / m; if (i >= ibase && i < ibase + myrows) T.As[(i - ibase) + c * 65] += V.aval[q]; }
            }
        } else if (G.selfasm) {
            for (int q = G.aq0 + tid; q < G.aq1; q += 256) { const int pos = V.apos[q]; const int i = pos % m, c = pos / m; if (i >= ibase && i < ibase + myrows) P[i + (size_t)c * ldp] += V.aval[q]; }
            __syncthreads();
        }
        const bool crit = staged && myq >= 0 && p == last_p;          // the link before my own pivot block: L21 of my rows goes to the panel AFTER my pivot block
        if (crit) { crit_Lr = T.Au; crit_Mp = Mp; crit_ibase = ibase; }
        chain_wait(&V.sflag_d[G.s], epoch, V.qstat + 1);              // L11, D, the pivot order (and the diagonal-block inverses) of link p are stored
        if (p == last_p) GSTAMP(1);
        if (tsp) tsp[4] = clock64();
        trsm_rows_impl<true>(V, Mp, T, ibase, myrows, tsp, staged != 0, crit ? 1 : 3);
        if (p == last_p) GSTAMP(2);
        if (myq >= 0) chain_signal(&V.sflag_s[4 * G.s + myq], epoch); // W(my rows, p) and L(my rows, p) are stored: the blocks after me may use them
        else __syncthreads();
        if (p == last_p) GSTAMP(3);
        if (tsp) tsp[5] = clock64();
        // my rows' entries in the columns of the later links:  T(i, c) -= sum_q L21(i, q) W21(c, q)
        {
            const double* Ds = T.Ds; const int* Ts = T.Ts; const double* As = T.As; const double* Lr = T.Au;
            auto lval = [&](const int r, const int q) -> double {
                if (Lr) return Lr[r + q * 65];                        // (kept by the panel solve)
                const int pt = Ts[q];
                const double wq = As[r + q * 65];
                if (pt == 1) return wq * Ds[q];
                if (pt == 2) return Ds[q] * wq + Ds[k + q] * As[r + (q + 1) * 65];
                return Ds[k + q - 1] * As[r + (q - 1) * 65] + Ds[q] * wq;
            };
            const int kp16 = T.kp16;
            const int rmax = (myq >= 0) ? myq : g - 1;
            // (a) the columns of OTHER links: W21 of their pivot rows comes from their row blocks (global), the target lives in L2
            if ((myq >= 0) ? (myq > p + 1) : (g - 1 > p)) {
                double bv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) { const int qq = 4 * u + l4; bv[u] = (qq < k) ? lval(16 * wave + l15, qq) : 0.0; }
                double* Tt = V.cb + G.t_off;
                const size_t ldt = (size_t)G.ldt;
                const int irow = (mystart - kend) + 16 * wave + l15;  // my row in T
                const bool rowok = 16 * wave + l15 < myrows;
                for (int r = p + 1; r <= rmax; ++r) {
                    if (r == myq) continue;
                    int cs = 0, cn = 0;                               // columns of link r, counted from the first column after link p
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (j == r) { cs = koff[j] - kend; cn = kk[j]; }
                    chain_wait(&V.sflag_s[4 * G.s + r], epoch, V.qstat + 1);
                    const double* Wg = V.wbuf + G.wb + k + cs;        // W21 rows of link r's pivots
                    const int nct = (cn + 15) >> 4;
                    for (int tc = 0; tc < nct; tc += 2) {
                        double a0[16], a1[16];
                        const int ra = 16 * tc + l15, rb = ra + 16;
#pragma unroll
                        for (int u = 0; u < 16; ++u) {
                            const int qq = 4 * u + l4;
                            a0[u] = (ra < cn && qq < k) ? Wg[ra + (size_t)qq * m] : 0.0;
                            a1[u] = (rb < cn && qq < k) ? Wg[rb + (size_t)qq * m] : 0.0;
                        }
                        double t0[4], t1[4];
#pragma unroll
                        for (int gg = 0; gg < 4; ++gg) {
                            const int c = 16 * tc + l4 + 4 * gg;
                            t0[gg] = (rowok && c < cn) ? Tt[irow + (size_t)(cs + c) * ldt] : 0.0;
                            t1[gg] = (rowok && c + 16 < cn) ? Tt[irow + (size_t)(cs + c + 16) * ldt] : 0.0;
                        }
                        v4f64 c0v = (v4f64){0.0, 0.0, 0.0, 0.0}, c1v = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int u = 0; u < 16; ++u)
                            if (4 * u < kp16) {
                                c0v = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], bv[u], c0v, 0, 0, 0);
                                c1v = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], bv[u], c1v, 0, 0, 0);
                            }
#pragma unroll
                        for (int gg = 0; gg < 4; ++gg) {
                            const int c = 16 * tc + l4 + 4 * gg;
                            if (rowok && c < cn) Tt[irow + (size_t)(cs + c) * ldt] = t0[gg] - c0v[gg];
                            if (rowok && c + 16 < cn) Tt[irow + (size_t)(cs + c + 16) * ldt] = t1[gg] - c1v[gg];
                        }
                    }
                }
            }
            if (tsp) tsp[6] = clock64();
            // (b) my OWN pivot block: both operands and the target are in LDS (lower 16 x 16 tiles, v_mfma_f64_16x16x4_f64; a lone
            //     wavefront issues one per ~150 cycles -- the same 1024 FMAs as 16 v_fmac_f64 take ~105, but those would want 8 LDS
            //     operands per step instead of 2: measured slower, tools/micro/fma_lds_latency.hip).  All 16 operand pairs of a tile
            //     are requested before its first MFMA.
            if (myq >= 0) {
                const int nt = (myrows + 15) >> 4;
                if (tsp) tsp[13] = clock64();
                int q = 0;
                for (int tc = 0; tc < nt; ++tc)
                    for (int ti = tc; ti < nt; ++ti, ++q) {
                        if ((q & 3) != wave) continue;
                        double av[16], bw[16];
                        v4f64 acc = (v4f64){0.0, 0.0, 0.0, 0.0};
                        if (Lr && kp16 == 64) {      // (straight-line: both LDS blocks are zero beyond column k)
#pragma unroll
                            for (int u = 0; u < 16; ++u) { av[u] = Lr[(16 * ti + l15) + (4 * u + l4) * 65]; bw[u] = As[(16 * tc + l15) + (4 * u + l4) * 65]; }
#pragma unroll
                            for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bw[u], acc, 0, 0, 0);
                        } else {
                            for (int u = 0; 4 * u < kp16; ++u) {
                                const int qq = 4 * u + l4;
                                acc = __builtin_amdgcn_mfma_f64_16x16x4f64((qq < k) ? lval(16 * ti + l15, qq) : 0.0, (qq < k) ? As[(16 * tc + l15) + qq * 65] : 0.0, acc, 0, 0, 0);
                            }
                        }
                        const int cc = 16 * tc + l15;
#pragma unroll
                        for (int gg = 0; gg < 4; ++gg) { const int rr = 16 * ti + l4 + 4 * gg; if (rr < myrows && cc < myrows && rr >= cc) Db[rr + cc * ldb] -= acc[gg]; }
                    }
                if (tsp) tsp[14] = clock64();
            }
        }
        __syncthreads();
        if (tsp) tsp[7] = clock64();
        if (p == last_p) GSTAMP(4);
    }
    if (myq < 0) return;
    {   // my own pivot block: a copy goes to the panel storage (the strict fall-back reads it there), then the factorisation in place
        double* Pq = V.L + Gq.panel_off;
        const size_t ldq = (size_t)Gq.ldp;
        const int i = tid & 63, cq = tid >> 6, kq = Gq.k;
#pragma unroll
        for (int e = 0; e < 16; ++e) { const int c = cq + 4 * e; if (i < kq && c < kq && i >= c) Pq[i + (size_t)c * ldq] = Db[i + c * ldb]; }
        FrontMeta Mq = M;
        Mq.s = Gq.s; Mq.c0 = Gq.c0; Mq.k = Gq.k; Mq.m = Gq.m; Mq.panel_off = Gq.panel_off; Mq.ldp = Gq.ldp; Mq.wb = Gq.wb; Mq.minv_off = Gq.minv_off;
        Mq.selfasm = 0; Mq.aq0 = Gq.aq0; Mq.aq1 = Gq.aq1; Mq.bigidx = Gq.bigidx;      // (the A entries are in already)
        __syncthreads();
        big_diag_body<4, 256, true>(V, Mq, smem_raw, &V.sflag_d[Gq.s], epoch, tsp, crit_Lr, crit_Mp, crit_ibase, myrows);
        if (tsp && tid == 0) { for (int q = 0; q < 28; ++q) V.dbg[64 + q] = tsp[q]; }
        GSTAMP(7);
    }
#undef GSTAMP
}



// ================================================================================================
// The 8-block primal-dual system on the device (reference IpPDFullSpaceSolver.cpp:377-664 SolveOnce,
// :666-793 ComputeResiduals, :795-820 ComputeResidualRatio).  A primal-dual vector is ONE array
//   [ x (nx) | s (ns) | y_c (nc) | y_d (nd) | z_L (nxl) | z_U (nxu) | v_L (nsl) | v_U (nsu) ];
// the bound-expansion matrices P are index lists, the iterate data (multipliers, slacks) live next to them, and W, J_c,
// J_d are the device-resident sources of the value assembly, read through a row view of those segments.
// ================================================================================================
struct PdView {
    int nx, ns, nc, nd, nxl, nxu, nsl, nsu;
    const int* ixl; const int* ixu; const int* isl; const int* isu;                 // positions of the bounded entries in x resp. s
    const double* zl; const double* zu; const double* vl; const double* vu;         // bound multipliers of the current iterate
    const double* sxl; const double* sxu; const double* ssl; const double* ssu;     // slacks of the current iterate
    const int* rptr; const int* rcol; const int* rslot;                             // row view of the W, J_c, J_d triplets (both triangles), slot order
    const double* tvals;                                                            // assembled triplet values (W_factor = 1: the plain W, J)
    unsigned long long* norms;                                                      // 3 order-preserving max accumulators
};
__device__ __forceinline__ int pd_off(const PdView& P, int blk)
{
    int o = 0;
    if (blk > 0) o += P.nx;  if (blk > 1) o += P.ns;  if (blk > 2) o += P.nc;  if (blk > 3) o += P.nd;
    if (blk > 4) o += P.nxl; if (blk > 5) o += P.nxu; if (blk > 6) o += P.nsl;
    return o;
}
// right-hand side of the augmented system (SolveOnce :418-424): the bound rows are eliminated into the x and s rows.
// pass 0: copy x | s | c | d;  pass 1: += P_L (rhs_zL / slack_L);  pass 2: -= P_U (rhs_zU / slack_U)   (the reference's order)
__global__ void k_pd_reduce(PdView P, const double* rhs, double* aug, int pass)
{
    const int n4 = P.nx + P.ns + P.nc + P.nd;
    const int tid0 = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    if (pass == 0) { for (int i = tid0; i < n4; i += nth) aug[i] = rhs[i]; return; }
    if (pass == 1) {
        const double* rz = rhs + pd_off(P, 4); const double* rv = rhs + pd_off(P, 6);
        for (int i = tid0; i < P.nxl; i += nth) aug[P.ixl[i]] += rz[i] / P.sxl[i];
        for (int i = tid0; i < P.nsl; i += nth) aug[P.nx + P.isl[i]] += rv[i] / P.ssl[i];
    } else {
        const double* rz = rhs + pd_off(P, 5); const double* rv = rhs + pd_off(P, 7);
        for (int i = tid0; i < P.nxu; i += nth) aug[P.ixu[i]] -= rz[i] / P.sxu[i];
        for (int i = tid0; i < P.nsu; i += nth) aug[P.nx + P.isu[i]] -= rv[i] / P.ssu[i];
    }
}
// back to eight blocks (SolveOnce :653-659): sol_z = S^{-1} (rhs_z -/+ Z P^T sol_x), then res = alpha sol + beta res
__device__ __forceinline__ double pd_combine(double alpha, double sol, double beta, double res)
{
    if (beta == 0.0) return (alpha == 1.0) ? sol : alpha * sol;
    if (beta == 1.0) return (alpha == 1.0) ? res + sol : ((alpha == -1.0) ? res - sol : res + alpha * sol);
    return alpha * sol + beta * res;
}
__global__ void k_pd_expand(PdView P, const double* rhs, const double* sol4, double* res, double alpha, double beta)
{
    const int n4 = P.nx + P.ns + P.nc + P.nd;
    const int tid0 = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    for (int i = tid0; i < n4; i += nth) res[i] = pd_combine(alpha, sol4[i], beta, res[i]);
    const int o4 = pd_off(P, 4), o5 = pd_off(P, 5), o6 = pd_off(P, 6), o7 = pd_off(P, 7);
    for (int i = tid0; i < P.nxl; i += nth) res[o4 + i] = pd_combine(alpha, (rhs[o4 + i] - P.zl[i] * sol4[P.ixl[i]]) / P.sxl[i], beta, res[o4 + i]);
    for (int i = tid0; i < P.nxu; i += nth) res[o5 + i] = pd_combine(alpha, (rhs[o5 + i] + P.zu[i] * sol4[P.ixu[i]]) / P.sxu[i], beta, res[o5 + i]);
    for (int i = tid0; i < P.nsl; i += nth) res[o6 + i] = pd_combine(alpha, (rhs[o6 + i] - P.vl[i] * sol4[P.nx + P.isl[i]]) / P.ssl[i], beta, res[o6 + i]);
    for (int i = tid0; i < P.nsu; i += nth) res[o7 + i] = pd_combine(alpha, (rhs[o7 + i] + P.vu[i] * sol4[P.nx + P.isu[i]]) / P.ssu[i], beta, res[o7 + i]);
}
__device__ __forceinline__ void pd_amax(unsigned long long* acc, double v)
{
    // |v| of a lane, maximum over the wavefront, one atomic per wavefront (non-negative doubles order like their bit patterns)
    double a = fabs(v);
    if (!(a == a)) a = __longlong_as_double(0x7ff0000000000000ll);     // NaN counts as +inf: the ratio test must fail loudly
    a = wave_max_all(a);
    if ((threadIdx.x & 63) == 0 && a > 0.0) atomicMax(acc, (unsigned long long)__double_as_longlong(a));
}
// residual of the UNREDUCED system (ComputeResiduals :666-793), rows x | s | c | d: one thread per row, entries in triplet order
//   resid_x = W res_x + J_c^T res_c + J_d^T res_d - P_xL res_zL + P_xU res_zU + delta_x res_x - rhs_x
//   resid_s = P_dU res_vU - P_dL res_vL - res_d - rhs_s + delta_s res_s
//   resid_c = J_c res_x - delta_c res_c - rhs_c          resid_d = J_d res_x - res_s - rhs_d - delta_d res_d
// (the P terms of the x and s rows are added by k_pd_resid_bounds, which also forms the four complementarity rows)
__global__ void k_pd_resid_rows(PdView P, const double* rhs, const double* res, double* resid, double dx, double ds, double dc, double dd)
{
    const int n4 = P.nx + P.ns + P.nc + P.nd;
    const int oS = P.nx, oC = P.nx + P.ns, oD = oC + P.nc;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < ((n4 + 63) & ~63); r += gridDim.x * blockDim.x) {
        double out = 0.0;
        if (r < n4) {
            double acc = 0.0;
            for (int q = P.rptr[r]; q < P.rptr[r + 1]; ++q) acc += P.tvals[P.rslot[q]] * res[P.rcol[q]];
            if (r < oS)      out = acc + dx * res[r] - rhs[r];
            else if (r < oC) { out = -res[oD + (r - oS)] - rhs[r]; if (ds != 0.0) out += ds * res[r]; }
            else if (r < oD) out = acc - dc * res[r] - rhs[r];
            else             { out = acc - res[oS + (r - oD)] - rhs[r]; if (dd != 0.0) out -= dd * res[r]; }
            resid[r] = out;
        }
    }
}
// pass 1: resid_x -= P_xL res_zL, resid_s -= P_dL res_vL and the lower complementarity rows; pass 2: the upper ones
//   resid_zL = Sl_xL res_zL + Z_L P_xL^T res_x - rhs_zL        resid_zU = Sl_xU res_zU - Z_U P_xU^T res_x - rhs_zU   (same for v / s)
__global__ void k_pd_resid_bounds(PdView P, const double* rhs, const double* res, double* resid, int pass)
{
    const int tid0 = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    if (pass == 1) {
        const int oz = pd_off(P, 4), ov = pd_off(P, 6);
        for (int i = tid0; i < P.nxl; i += nth) { const int j = P.ixl[i]; resid[j] -= res[oz + i]; resid[oz + i] = res[oz + i] * P.sxl[i] + res[j] * P.zl[i] - rhs[oz + i]; }
        for (int i = tid0; i < P.nsl; i += nth) { const int j = P.nx + P.isl[i]; resid[j] -= res[ov + i]; resid[ov + i] = res[ov + i] * P.ssl[i] + res[j] * P.vl[i] - rhs[ov + i]; }
    } else {
        const int oz = pd_off(P, 5), ov = pd_off(P, 7);
        for (int i = tid0; i < P.nxu; i += nth) { const int j = P.ixu[i]; resid[j] += res[oz + i]; resid[oz + i] = res[oz + i] * P.sxu[i] - res[j] * P.zu[i] - rhs[oz + i]; }
        for (int i = tid0; i < P.nsu; i += nth) { const int j = P.nx + P.isu[i]; resid[j] += res[ov + i]; resid[ov + i] = res[ov + i] * P.ssu[i] - res[j] * P.vu[i] - rhs[ov + i]; }
    }
}
// max norms of rhs, res and resid over all eight blocks (ComputeResidualRatio :795-820)
__global__ void k_pd_norms(PdView P, const double* rhs, const double* res, const double* resid, long long len8)
{
    const long long lenp = (len8 + 63) & ~63ll;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < lenp; i += (long long)gridDim.x * blockDim.x) {
        const bool v = i < len8;
        pd_amax(P.norms + 0, v ? rhs[i] : 0.0); pd_amax(P.norms + 1, v ? res[i] : 0.0); pd_amax(P.norms + 2, v ? resid[i] : 0.0);
    }
}

// ================================================================================================
// multi-GPU pieces (one process per GPU, subtrees sharded, top of the tree replicated)
// ================================================================================================
// Every rank adds what IT knows about each replicated (top) front into that front's m x m arena square: rank 0 the
// A entries, every rank the contribution blocks of its own subtree roots.  One wavefront per front column, children
// in fixed order => deterministic.  The arena is then summed over ranks (RCCL all-reduce) by the caller.
__global__ __launch_bounds__(256) void k_arena_assemble(DevView V, int list_off, int who)
{
    const FrontMeta M = V.fmeta[list_off + blockIdx.y];
    const int s = M.s, c0 = M.c0, k = M.k, r0 = M.r0, m = M.m; (void)s; (void)c0; (void)r0; (void)k;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fc = blockIdx.x * 4 + wave;
    const bool active = fc < m;
    double* col = active ? V.arena + V.arena_off[s] + (size_t)fc * m : nullptr;
    for (int cp = M.ch0; cp < M.ch1; ++cp) {
        const ChildMeta Cm = V.cmeta[cp];
        const int ch = Cm.ch; (void)ch;
        if (active && Cm.owner == who) {
            const int mc = Cm.mc;
            const int* relc = V.rel + Cm.relbase;
            int lo = 0, hi = mc;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (relc[mid] < fc) lo = mid + 1; else hi = mid; }
            if (lo < mc && relc[lo] == fc) {
                const double* C = V.cb + Cm.cb_off + (size_t)lo * Cm.ldt;
                for (int a = lo + lane; a < mc; a += 64) col[relc[a]] += C[a];
            }
        }
        __syncthreads();
    }
}
// forward-solve contributions of the children this rank reports (code `who`, see setup) to the replicated fronts above them, ADDED to the
// accumulators the caller zeroed at the start of the solve (and sums over the ranks afterwards)
__global__ __launch_bounds__(256) void k_top_rhs_assemble(DevView V, int list_off, int who)
{
    const FrontMeta M = V.fmeta[list_off + blockIdx.x];
    const int s = M.s, c0 = M.c0, k = M.k, r0 = M.r0, m = M.m; (void)s; (void)c0; (void)r0; (void)k; (void)m;
    double* tr = V.top_rhs + V.top_rhs_off[s];
    for (int cp = M.ch0; cp < M.ch1; ++cp) {
        const ChildMeta Cm = V.cmeta[cp];
        const int ch = Cm.ch; (void)ch;
        if (Cm.owner == who) {
            const int base = Cm.relbase, mc = Cm.mc;
            for (int t = threadIdx.x; t < mc; t += 256) tr[V.rel[base + t]] += V.cvec[Cm.cvbase + t];
        }
        __syncthreads();
    }
}
// this rank's part of the solution (own subtrees + the replicated columns it reports: col_owner is this rank's view, setup), zero
// elsewhere => the caller's all-reduce(sum) assembles the full vector on every rank
__global__ void k_store_sol_mg(DevView V, double* b)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < V.n; i += gridDim.x * blockDim.x) {
        const int o = V.col_owner[i];
        b[V.perm[i]] = o == V.rank ? V.scale[i] * V.xw[i] : 0.0;
    }
}

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
    bool scale_identity = true;
    double* d_user_scale = nullptr;      // caller-supplied scaling factors, original numbering (scaling mode 2)
    // scaling mode at run time: 0 none, 1 Ruiz on device, 2 the caller's factors (MA97 semantics, IpMa97SolverInterface.cpp:641-678)
    bool set_scaling(int mode, const double* user) {
        DeviceGuard guard(dev);
        if (!ready) { err_ = "set_scaling: solver not set up"; return false; }
        if (mode < 0 || mode > 3 || (mode == 2 && !user)) { err_ = "set_scaling: mode 0 (none), 1 (ruiz), 2 (user factors, non-null) or 3 (matching)"; return false; }
        if (mode == 2) {
            if (!d_user_scale) { HIPCHK(hipMalloc((void**)&d_user_scale, std::max<size_t>(S->n, 1) * sizeof(double))); allocs.push_back(d_user_scale); }
            HIPCHK(hipMemcpyAsync(d_user_scale, user, (size_t)S->n * sizeof(double), hipMemcpyHostToDevice, stream));
            HIPCHK(hipStreamSynchronize(stream));      // `user` is the caller's pageable memory
        }
        if (mode == 3 && !d_user_scale) { HIPCHK(hipMalloc((void**)&d_user_scale, std::max<size_t>(S->n, 1) * sizeof(double))); allocs.push_back(d_user_scale); }
        if (mode != opt.scaling && g_factor) { (void)hipGraphExecDestroy(g_factor); g_factor = nullptr; }     // the captured sequence differs
        opt.scaling = mode;
        return true;
    }
    // scaling mode 3: maximum-product matching scaling (MC64-style, matching_scaling.cpp) of the values now in V.tvals.  Host
    // algorithm, like the analysis: gather on the device, one D2H of the nnz(A) summed values, the matching, one H2D of n
    // factors -- which the factorisation then applies exactly like caller-supplied ones.
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
        if (opt.verbose) fprintf(stderr, "[mi355x_kkt] matching scaling: %d unmatched columns\n", unmatched);
        HIPCHK(hipMemcpyAsync(d_user_scale, so.data(), (size_t)Sy.n * sizeof(double), hipMemcpyHostToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream));          // `so` is about to go out of scope
        return true;
    }
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
    struct GrpSched { std::vector<int> g0, g1, split, nrb, tiles64, tiles, la1, la2; std::vector<hipEvent_t> evA, evB; };
    GrpSched gs_single, gs_local;              // one-GPU schedule; multi-GPU: the rank's own subtrees
    std::vector<GrpSched> gs_stage;            // multi-GPU: the replicated fronts this rank holds, per exchange step (sn_gdepth)
    GrpSched* gs_cur = nullptr;                // ... the step launch_fronts is working on
    std::vector<hipEvent_t> la_evA, la_evB;
    // chain look-ahead (single-GPU schedule, levels whose fronts are all pure in-place chain links): the critical path
    //   pivot block (k_big_diag_reg) -> first row block of the panel (k_big_trsm, 1 workgroup) -> the NEXT link's 64 x 64 pivot
    //   block (k_big_schur64, tile (0,0))
    // stays on the main stream; the bulk of the panel solve and of the trailing update trails on `stream3` (the far part of
    // a split group-end update on `stream2`), overlapped with the next link's latency-bound pivot block
    hipStream_t stream3 = nullptr;
    std::vector<char> lv_chain; bool chain_la = true; int chain_maxf = 64;
    std::vector<hipEvent_t> chD, chLA, chN, chG1, chFar;
    hipEvent_t ch_bulk_last = nullptr, ch_far_last = nullptr; bool ch_bulk_pending = false, ch_far_pending = false;
    hipStream_t stream2 = nullptr; bool la_pending = false; hipEvent_t la_last = nullptr; bool lookahead = true, la_any = false; int la_wgs = 1 << 20, la_min_nt = 12;
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
        return true;
    }
    bool factor_assembled(const double* scale, const double* shift, FactorStats& st) {
        {
            DeviceGuard guard(dev);
            if (!ready || asm_.nseg == 0) { err_ = "factor_assembled: assembly_define first"; return false; }
            long long mx = 1;
            for (int q = 0; q < asm_.nseg; ++q) { asm_.scale[q] = scale[q]; asm_.shift[q] = shift[q]; mx = std::max(mx, asm_.len[q]); }
            hipLaunchKernelGGL(k_assemble_segments, dim3(grid1d(mx), asm_.nseg), dim3(256), 0, stream, (double*)V.tvals, asm_);
            HIPCHK(hipGetLastError());
            have_values = true;
        }
        return factor(nullptr, true, st);       // the values are on the device: the "refactor" path, no host buffer involved
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
        const char* (*GetErrorString)(ncclResult_t) = nullptr;
    } rccl;
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
        R.GetErrorString = (decltype(R.GetErrorString))dlsym(R.lib, "ncclGetErrorString");
        if (!R.GetUniqueId || !R.CommInitRank || !R.AllReduce || !R.CommDestroy) { err = "librccl.so lacks the nccl* entry points"; return false; }
        return true;
    }
    bool set_comm_rccl(const void* id128) {
        DeviceGuard guard(dev);
        if (!ready || !multi) { err_ = "set_comm_rccl: not a multi-GPU handle (nranks > 1 at create, analyse first)"; return false; }
        if (!rccl_load(rccl, err_)) return false;
        if (rccl.comm) { (void)rccl.CommDestroy(rccl.comm); rccl.comm = nullptr; }
        ncclUniqueId id; std::memcpy(&id, id128, sizeof(id));
        ncclResult_t r = rccl.CommInitRank(&rccl.comm, opt.nranks, id, opt.rank);
        if (r != ncclSuccess) { err_ = std::string("ncclCommInitRank: ") + (rccl.GetErrorString ? rccl.GetErrorString(r) : "error"); rccl.comm = nullptr; return false; }
        comm_kind = 2; return true;
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
    void prof_begin(int kind) {
        if (!prof_on) return;
        if (prof_used + 2 > prof_ev.size()) { size_t old = prof_ev.size(); prof_ev.resize(old + 64); for (size_t i = old; i < prof_ev.size(); ++i) (void)hipEventCreate(&prof_ev[i]); }
        (void)hipEventRecord(prof_ev[prof_used], stream); prof_kind.push_back(kind);
    }
    void prof_end() { if (!prof_on) return; (void)hipEventRecord(prof_ev[prof_used + 1], stream); prof_used += 2; }
    void prof_collect() {
        (void)hipStreamSynchronize(stream);
        for (size_t i = 0; i < prof_used; i += 2) { float ms = 0; (void)hipEventElapsedTime(&ms, prof_ev[i], prof_ev[i + 1]); int kd = prof_kind[i / 2]; prof_ms[kd] += ms; prof_launches[kd]++; }
        prof_used = 0; prof_kind.clear();
    }
    ~NumericImpl() { release(); for (auto e : prof_ev) (void)hipEventDestroy(e); }
    void release() {
        DeviceGuard guard(dev);
        if (rccl.comm && rccl.CommDestroy) { (void)rccl.CommDestroy(rccl.comm); rccl.comm = nullptr; }
        comm_kind = 0;
        if (g_factor) { (void)hipGraphExecDestroy(g_factor); g_factor = nullptr; }
        if (g_solve) { (void)hipGraphExecDestroy(g_solve); g_solve = nullptr; }
        for (void* p : allocs) (void)hipFree(p);
        allocs.clear();
        if (d_rhs) { (void)hipFree(d_rhs); d_rhs = nullptr; d_rhs_cap = 0; }
        if (asm_pool) { (void)hipFree(asm_pool); asm_pool = nullptr; }
        if (asm_hpool) { (void)hipHostFree(asm_hpool); asm_hpool = nullptr; }
        asm_.nseg = 0;
        pd_free();
        if (h_vals) { (void)hipHostFree(h_vals); h_vals = nullptr; }
        if (h_stats) { (void)hipHostFree(h_stats); h_stats = nullptr; }
        if (ev0) { (void)hipEventDestroy(ev0); ev0 = nullptr; }
        if (ev1) { (void)hipEventDestroy(ev1); ev1 = nullptr; }
        for (auto e : la_evA) if (e) (void)hipEventDestroy(e);
        for (auto e : la_evB) if (e) (void)hipEventDestroy(e);
        la_evA.clear(); la_evB.clear();
        { std::vector<GrpSched*> all{&gs_single, &gs_local}; for (auto& g : gs_stage) all.push_back(&g);
          for (GrpSched* g : all) { for (auto e : g->evA) if (e) (void)hipEventDestroy(e); for (auto e : g->evB) if (e) (void)hipEventDestroy(e); g->evA.clear(); g->evB.clear(); } }
        for (auto* v : {&chD, &chLA, &chN, &chG1, &chFar}) { for (auto e : *v) if (e) (void)hipEventDestroy(e); v->clear(); }
        if (stream3) { (void)hipStreamDestroy(stream3); stream3 = nullptr; }
        if (stream2) { (void)hipStreamDestroy(stream2); stream2 = nullptr; }
        if (stream) { (void)hipStreamDestroy(stream); stream = nullptr; }
        ready = false;
    }
    template <class T> bool upload(const std::vector<T>& h, const T** d) {
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

    bool setup(const Symbolic& Sy, const NumericOptions& o) {
        release(); S = &Sy; opt = o;
        auto now_ = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double t_prev = now_();
        auto lap = [&](const char* what) { if (opt.verbose >= 2) { const double t = now_(); fprintf(stderr, "[mi355x_kkt]   setup %-28s %.3f s\n", what, t - t_prev); t_prev = t; } };
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
            err_ = "no HIP device available: the MI355X KKT backend has no CPU fallback"; have_device = false; return false; }
        if (opt.device >= 0) dev = opt.device; else HIPCHK(hipGetDevice(&dev));
        DeviceGuard guard(dev);
        have_device = true;
        {   // the main stream carries the latency-bound pivot chains: highest priority; the look-ahead stream the lowest
            int plo = 0, phi = 0;
            (void)hipDeviceGetStreamPriorityRange(&plo, &phi);
            HIPCHK(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, phi));
            HIPCHK(hipStreamCreateWithPriority(&stream2, hipStreamNonBlocking, plo));
            HIPCHK(hipStreamCreateWithPriority(&stream3, hipStreamNonBlocking, (plo + phi) / 2));
        }
        lookahead = getenv("MI355X_KKT_NO_LOOKAHEAD") == nullptr;
        chain_la = getenv("MI355X_KKT_CHAIN_LA") != nullptr;      // default OFF: measured slower (DESIGN.md "measured design decisions")
        if (const char* e = getenv("MI355X_KKT_CHAIN_LA_MAXF")) chain_maxf = std::max(1, atoi(e));
        if (const char* e = getenv("MI355X_KKT_LA_WGS")) la_wgs = std::max(1, atoi(e));          // development knobs
        if (const char* e = getenv("MI355X_KKT_LA_MIN_NT")) la_min_nt = std::max(3, atoi(e));
        HIPCHK(hipEventCreate(&ev0)); HIPCHK(hipEventCreate(&ev1));
        if (opt.prewarmed_vals && opt.prewarmed_count >= std::max<size_t>(Sy.nnz_in, 1)) h_vals = (double*)opt.prewarmed_vals;       // (made while the analysis ran)
        else {
            if (opt.prewarmed_vals) (void)hipHostFree(opt.prewarmed_vals);
            HIPCHK(hipHostMalloc((void**)&h_vals, std::max<size_t>(Sy.nnz_in, 1) * sizeof(double), hipHostMallocDefault));
        }
        opt.prewarmed_vals = nullptr;
        HIPCHK(hipHostMalloc((void**)&h_stats, 8 * sizeof(int), hipHostMallocDefault));
        lap("device, streams, pinned buffer");
        std::vector<long long> poff(Sy.panel_off.begin(), Sy.panel_off.end()), coff(Sy.cb_off.begin(), Sy.cb_off.end()), moff(Sy.minv_off.begin(), Sy.minv_off.end());
        multi = opt.nranks > 1 || getenv("MI355X_KKT_FORCE_MULTI") != nullptr;   // (1-rank multi path: plumbing tests on a 1-GPU box)
        std::vector<int> lvl_list(Sy.level_sn);
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
            // top-rhs accumulators for every replicated front; arena squares only for those with a child from outside their range (the joins):
            // that is all the all-reduce has to carry (A is replicated input, not reduced).  Both laid out step by step, the same on every rank.
            std::vector<char> is_join(Sy.num_sn, 0);
            for (int c = 0; c < Sy.num_sn; ++c) if (crosses(c)) is_join[Sy.sn_parent[c]] = 1;
            abeg.assign(ndepth, 0); aend.assign(ndepth, 0); tbeg.assign(ndepth, 0); tend.assign(ndepth, 0);
            for (int d = 0; d < ndepth; ++d) {
                abeg[d] = arena_doubles; tbeg[d] = toprhs_doubles;
                for (int s = 0; s < Sy.num_sn; ++s) if (Sy.sn_owner[s] < 0 && Sy.sn_gdepth[s] == d) {
                    const long long m = Sy.sn_rowptr[s + 1] - Sy.sn_rowptr[s];
                    troff[s] = toprhs_doubles; toprhs_doubles += m;
                    if (is_join[s]) { aoff[s] = arena_doubles; arena_doubles += m * m; }
                }
                aend[d] = arena_doubles; tend[d] = toprhs_doubles;
            }
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
        chain_segs.clear(); seg_at_lv0.assign(Sy.num_levels, -1); seg_at_lv1.assign(Sy.num_levels, -1);
        chain_solve = getenv("MI355X_KKT_NO_CHAIN_SOLVE") == nullptr;
        fuse_dt = getenv("MI355X_KKT_NO_FUSE_DT") == nullptr;
        V.fastpiv = getenv("MI355X_KKT_NO_FASTPIV") == nullptr ? 1 : 0;
        V.asm_pull = getenv("MI355X_KKT_NO_ASM_PULL") == nullptr ? 1 : 0;
        V.fastu = 1e-4; if (const char* e = getenv("MI355X_KKT_FASTPIV_FLOOR")) V.fastu = atof(e);      // (0.01 up to r03a: 9 % of the synth_1e6 blocks then took the strict loop and set the pace of their level: 23.1 -> 22.0 ms)
        if (const char* e = getenv("MI355X_KKT_FUSE_DT_MAXWG")) fuse_dt_maxwg = atoi(e);
        pair_solve = getenv("MI355X_KKT_NO_PAIR_SOLVE") == nullptr && !multi;
        wave_kmax.assign(Sy.num_levels, 0); wave_mmax.assign(Sy.num_levels, 0); wave_mmin.assign(Sy.num_levels, 1 << 30);
        for (int sn = 0; sn < Sy.num_sn; ++sn) if (Sy.sn_class[sn] == FC_WAVE) {
            wave_kmax[Sy.sn_level[sn]] = std::max(wave_kmax[Sy.sn_level[sn]], Sy.sn_colptr[sn + 1] - Sy.sn_colptr[sn]);
            wave_mmax[Sy.sn_level[sn]] = std::max(wave_mmax[Sy.sn_level[sn]], Sy.sn_rowptr[sn + 1] - Sy.sn_rowptr[sn]);
            wave_mmin[Sy.sn_level[sn]] = std::min(wave_mmin[Sy.sn_level[sn]], Sy.sn_rowptr[sn + 1] - Sy.sn_rowptr[sn]);
        }
        if (const char* e = getenv("MI355X_KKT_CHAIN_SOLVE_MAXC")) chain_maxc = std::max(1, atoi(e));
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
                            ChainLink L{}; L.panel_off = Sy.panel_off[sn]; L.minv_off = Sy.minv_off[sn]; L.c0 = Sy.sn_colptr[sn]; L.k = Kc(sn); L.ldp = Sy.sn_ldp[sn];
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
        if (getenv("MI355X_KKT_SOLVE_TRACE") && !wgf.empty()) {
            strace_n = 4 * (wgf.size() + wgb.size());
            if (!dalloc(&V.strace, strace_n)) return false;
            strace_desc.clear();
            for (size_t i = 0; i < wgf.size(); ++i) { const ChainDesc& D = chd[wgf[i]]; strace_desc.push_back({wgf[i], (int)i - D.wg0f - chain_segs[0].wgf0 * 0, D.nlinks, D.tail}); }
            for (size_t i = 0; i < wgb.size(); ++i) { const ChainDesc& D = chd[wgb[i]]; strace_desc.push_back({wgb[i], (int)i - D.wg0b - D.nlinks * ((D.tail + 255) / 256), D.nlinks, D.tail}); }      // (dot workgroups: negative)
        }
        if (!dalloc(&V.sflag_dot, (size_t)std::max(ndots, 1) * FLAG_STRIDE) || !dalloc(&V.dpart, (size_t)std::max(ndots, 1) * 64) ||
            !dalloc(&V.sflag_t, (size_t)std::max(ntailflags, 1) * FLAG_STRIDE) || !dalloc(&V.sflag_b, std::max<size_t>(chl.size(), 1) * FLAG_STRIDE)) return false;
        {   // tagged solution entries of the segments' columns (indexed by column; + 64: a wavefront polls 64 entries from a link's first column)
            const size_t nt = chl.empty() ? 1 : (size_t)Sy.n + 64;
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
                G.panel_off = Sy.panel_off[l]; G.wb = Sy.wb_off[l]; G.minv_off = Sy.minv_off[l]; G.cv = Sy.cv_off[l]; G.tr = troff[l];
                G.c0 = Sy.sn_colptr[l]; G.k = Sy.sn_colptr[l + 1] - G.c0; G.r0 = Sy.sn_rowptr[l]; G.m = Sy.sn_rowptr[l + 1] - G.r0;
                G.ldp = Sy.sn_ldp[l]; G.ch0 = Sy.child_ptr[l]; G.ch1 = Sy.child_ptr[l + 1]; G.alias = Sy.alias_child[l] >= 0 ? 1 : 0;
                G.t_off = Sy.cb_off[l]; G.ldt = Sy.sn_ldt[l]; G.s = l; G.aq0 = Sy.acolptr[G.c0]; G.aq1 = Sy.acolptr[G.c0 + G.k]; G.bigidx = bigidx_of[l];
                G.selfasm = ((!multi || aoff[l] < 0) && getenv("MI355X_KKT_NO_SELFASM") == nullptr && Sy.alias_child[l] >= 0 && Sy.child_ptr[l + 1] - Sy.child_ptr[l] == 1) ? 1 : 0;
                gt.push_back(G); gcols_of[sn] += G.k;
            }
        }
        // look-ahead candidates: group-last BIG fronts (>= 12 tile rows) whose chain continues with a PURE next group (links
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
            const long long la_gate = getenv("MI355X_KKT_LA_MIN_TILES") ? atoll(getenv("MI355X_KKT_LA_MIN_TILES")) : 4000;     // (tests force 0)
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
            const bool xcd_aware = getenv("MI355X_KKT_NO_XCD_TILES") == nullptr;
            for (int sn = 0; sn < Sy.num_sn; ++sn) if (xcd_aware && Sy.sn_class[sn] == FC_BIG && Sy.grp_rem[sn] == 0) {
                const int mu = (Sy.sn_rowptr[sn + 1] - Sy.sn_rowptr[sn]) - (Sy.sn_colptr[sn + 1] - Sy.sn_colptr[sn]);
                const int nt = (mu + 127) / 128;
                if (nt < 12) continue;
                ttab_of[sn] = table_for(nt);
                if (split_of[sn]) ttab2_of[sn] = table_for(nt - 2);
            }
        }
        if (!upload(tile_tab, &V.tile_tab)) return false;
        const bool selfasm_on = getenv("MI355X_KKT_NO_SELFASM") == nullptr;
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
        if (!multi && getenv("MI355X_KKT_NO_FUSE_UPD") == nullptr)
            for (int lv = 0; lv < Sy.num_levels; ++lv) {
                bool all = true; int cnt = 0, tl = 0;
                for (int q = Sy.level_ptr[(size_t)lv * FC_COUNT + FC_BIG]; q < Sy.level_ptr[(size_t)lv * FC_COUNT + FC_BIG + 1]; ++q) {
                    const int sn = Sy.level_sn[q]; ++cnt;
                    if (Sy.grp_rem[sn] <= 0) all = false;
                    tl = std::max(tl, schur_tiles64(Sy, sn));
                }
                lv_narrow_tiles[lv] = (cnt > 0 && all) ? tl : 0;
            }
        lv_chain.assign(Sy.num_levels, 0);
        if (!multi && selfasm_on && chain_la)
            for (int lv = 0; lv < Sy.num_levels; ++lv) {
                const int nbig = Sy.level_ptr[(size_t)lv * FC_COUNT + FC_BIG + 1] - Sy.level_ptr[(size_t)lv * FC_COUNT + FC_BIG];
                const int nall = Sy.level_ptr[(size_t)lv * FC_COUNT + FC_COUNT] - Sy.level_ptr[(size_t)lv * FC_COUNT];
                int kmax = 0;
                for (int q = Sy.level_ptr[(size_t)lv * FC_COUNT + FC_BIG]; q < Sy.level_ptr[(size_t)lv * FC_COUNT + FC_BIG + 1]; ++q) kmax = std::max(kmax, Sy.sn_colptr[Sy.level_sn[q] + 1] - Sy.sn_colptr[Sy.level_sn[q]]);
                lv_chain[lv] = (lv_asm_skip[lv] && nbig == nall && nbig <= chain_maxf && kmax <= 64) ? 1 : 0;
            }
        {
            int nchain = 0;
            for (int lv = 0; lv < Sy.num_levels; ++lv) nchain += lv_chain[lv];
            if (nchain < 8) std::fill(lv_chain.begin(), lv_chain.end(), 0);      // not worth leaving the graph replay for
            else la_any = true;                                                  // multi-stream schedule => eager launches (see factor())
            chD.assign(Sy.num_levels, nullptr); chLA.assign(Sy.num_levels, nullptr); chN.assign(Sy.num_levels, nullptr); chG1.assign(Sy.num_levels, nullptr); chFar.assign(Sy.num_levels, nullptr);
            for (int lv = 0; lv < Sy.num_levels; ++lv) if (lv_chain[lv])
                for (auto* v : {&chD, &chLA, &chN, &chG1, &chFar}) HIPCHK(hipEventCreateWithFlags(&(*v)[lv], hipEventDisableTiming));
            if (opt.verbose) fprintf(stderr, "[mi355x_kkt] chain look-ahead on %d of %d levels\n", nchain >= 8 ? nchain : 0, Sy.num_levels);
        }
        // ---- grouped schedule: every chain group is factored at the level of its first link (k_grp_fused + update) ----
        grouped = Sy.maxsupernode <= 64 && selfasm_on && getenv("MI355X_KKT_NO_GROUPED") == nullptr;
        {
            auto order_of = [&](int sn) { return Sy.sn_rowptr[sn + 1] - Sy.sn_rowptr[sn]; };
            auto cols_of = [&](int sn) { return Sy.sn_colptr[sn + 1] - Sy.sn_colptr[sn]; };
            // which: 0 = every front (one GPU), 1 = the rank's own subtrees, 2 + d = the replicated fronts of exchange step d held by this rank (an
            // in-place chain never crosses an ownership boundary -- symbolic.cpp only aliases fronts of one owner and one range of ranks -- so neither does a group)
            auto build_groups = [&](GrpSched& G, int which) -> bool {
                for (auto* v : {&G.g0, &G.g1, &G.split, &G.nrb, &G.tiles64, &G.tiles, &G.la1, &G.la2}) v->assign(Sy.num_levels, 0);
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
        std::vector<FrontMeta> fm(lvl_list.size());
        for (size_t q = 0; q < lvl_list.size(); ++q) {
            const int sn = lvl_list[q];
            FrontMeta& M = fm[q];
            M.s = sn; M.c0 = Sy.sn_colptr[sn]; M.k = Sy.sn_colptr[sn + 1] - M.c0; M.r0 = Sy.sn_rowptr[sn]; M.m = Sy.sn_rowptr[sn + 1] - M.r0;
            M.aq0 = Sy.acolptr[M.c0]; M.aq1 = Sy.acolptr[M.c0 + M.k]; M.ch0 = Sy.child_ptr[sn]; M.ch1 = Sy.child_ptr[sn + 1]; M.alias = Sy.alias_child[sn] >= 0 ? 1 : 0;
            M.ldp = Sy.sn_ldp[sn]; M.ldt = Sy.sn_ldt[sn];
            M.panel_off = Sy.panel_off[sn]; M.cb_off = Sy.cb_off[sn]; M.minv_off = Sy.minv_off[sn];
            M.cv = Sy.cv_off[sn]; M.wb = Sy.wb_off[sn]; M.gpart = Sy.gpart_off[sn];
            M.gbase = gbase_of[sn]; M.gpos = Sy.grp_pos[sn]; M.grem = Sy.grp_rem[sn]; M.gcols = gcols_of[sn]; M.split = split_of[sn]; M.ttab = ttab_of[sn]; M.ttab2 = ttab2_of[sn];
            // (multi-GPU: not for a front at a subtree join -- its square comes out of the all-reduced arena)
            M.selfasm = ((!multi || aoff[sn] < 0) && selfasm_on && Sy.sn_class[sn] == FC_BIG && Sy.alias_child[sn] >= 0 && Sy.child_ptr[sn + 1] - Sy.child_ptr[sn] == 1) ? 1 : 0; M.bigidx = bigidx_of[sn];
            {   // 1: in-place chain link whose only child is the chain child, 2: no children at all => the fused forward kernel applies
                const int nch = Sy.child_ptr[sn + 1] - Sy.child_ptr[sn];
                M.solo = (Sy.alias_child[sn] >= 0 && nch == 1) ? 1 : ((Sy.alias_child[sn] < 0 && nch == 0) ? 2 : 0);
            }
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
            cm[q].owner = child_code(ch); cm[q].cb_off = Sy.cb_off[ch]; cm[q].ldt = Sy.sn_ldt[ch]; cm[q].aliased = 0; cm[q].cvbase = Sy.cv_off[ch] + kc;
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
        // the inertia / pivot counts are summed over the ranks: a replicated front is counted by the first rank of its range (-1 in this rank's view), -3 = not here
        std::vector<int> stat_owner(Sy.sn_owner);
        if (multi) for (int sn = 0; sn < Sy.num_sn; ++sn) if (Sy.sn_owner[sn] < 0) stat_owner[sn] = Sy.sn_glo[sn] == opt.rank ? -1 : -3;
        if (!upload(Sy.sn_colptr, &V.sn_colptr) || !upload(Sy.sn_rowptr, &V.sn_rowptr) || !upload(Sy.sn_rows, &V.sn_rows) ||
            !upload(Sy.rel, &V.rel) || !upload(Sy.child_ptr, &V.child_ptr) || !upload(Sy.child_idx, &V.child_idx) ||
            !upload(stat_owner, &V.sn_owner) || !upload(poff, &V.panel_off) || !upload(coff, &V.cb_off) || !upload(moff, &V.minv_off) ||
            !upload(Sy.acolptr, &V.acolptr) || !upload(Sy.apos, &V.apos) || !upload(Sy.arow, &V.arow) || !upload(Sy.acol, &V.acol) ||
            !upload(Sy.dup_ptr, &V.dup_ptr) || !upload(Sy.dup_src, &V.dup_src) || !upload(Sy.rslot_ptr, &V.rslot_ptr) || !upload(Sy.rslot_idx, &V.rslot_idx) || !upload(Sy.rslot_col, &V.rslot_col) || !upload(lvl_list, &V.level_sn) ||
            !upload(Sy.sn_parent, &V.sn_parent) || !upload(colown, &V.col_owner) || !upload(aoff, &V.arena_off) || !upload(troff, &V.top_rhs_off) ||
            !upload(Sy.perm, &V.perm)) return false;
        lap("uploads");
        double* tv = nullptr;
        if (!dalloc(&tv, Sy.nnz_in)) return false; V.tvals = tv;
        V.rslot_len = (int)Sy.rslot_idx.size();
        if (!dalloc(&V.arv, Sy.rslot_idx.size()) || !dalloc(&V.aval, Sy.nnz_a) || !dalloc(&V.scale, Sy.n) || !dalloc(&V.scale2, Sy.n) || !dalloc(&V.rowmax, Sy.n) ||
            !dalloc(&V.L, (size_t)(Sy.l_doubles + Sy.cb_doubles)) || !dalloc(&V.wbuf, (size_t)Sy.wbuf_doubles) || !dalloc(&V.minv, (size_t)Sy.minv_doubles) ||
            !dalloc(&V.dinv, Sy.n) || !dalloc(&V.doff, Sy.n) || !dalloc(&V.ptype, Sy.n) || !dalloc(&V.lperm, Sy.n) ||
            !dalloc(&V.fstat, Sy.num_sn) || !dalloc(&V.xw, Sy.n) || !dalloc(&V.ybuf, Sy.n) || !dalloc(&V.zb, Sy.n) || !dalloc(&V.bw, Sy.n) || !dalloc(&V.xacc, Sy.n) || !dalloc(&V.cvec, (size_t)Sy.cvec_doubles) || !dalloc(&V.gpart, (size_t)Sy.gpart_doubles) ||
            !dalloc(&d_stats, 8) || !dalloc(&V.colfail, Sy.n) || !dalloc(&V.zpiv, Sy.n) || !dalloc(&V.cnorm, Sy.n) ||
            !dalloc(&V.sflag_d, Sy.num_sn) || !dalloc(&V.sflag_s, 4 * (size_t)Sy.num_sn) || !dalloc(&V.tcnt, Sy.num_sn) || !dalloc(&V.sepoch, 4)) return false;
        V.qstat = d_stats + 4;
        if (opt.scaling == 3) { HIPCHK(hipMalloc((void**)&d_user_scale, std::max<size_t>(Sy.n, 1) * sizeof(double))); allocs.push_back(d_user_scale); }
        else if (opt.scaling == 2) opt.scaling = 1;       // (user factors can only come through set_scaling)
        V.cb = V.L + Sy.l_doubles;          // one pool: panels of in-place chain fronts live inside the cb part
        V.arena = nullptr; V.top_rhs = nullptr; V.rank = opt.rank; V.dbg = nullptr;
        if (getenv("MI355X_KKT_DEBUG_CLOCKS")) { if (!dalloc(&V.dbg, 128)) return false; }
        if (multi) { if (!dalloc(&V.arena, (size_t)arena_doubles) || !dalloc(&V.top_rhs, (size_t)toprhs_doubles)) return false; }
        V.pivtol = opt.pivtol; V.pivtol2 = std::max(opt.pivtol, opt.pivtolmax); V.small = opt.small; V.n = Sy.n; V.nnz_a = Sy.nnz_a; V.nsn = Sy.num_sn;
        // allow the large dynamic LDS sizes
        HIPCHK(hipFuncSetAttribute((const void*)k_big_diag_reg<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)(k_big_diag_reg<4, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)(k_front_reg<64, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)(k_front_reg<64, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)(k_front_reg<64, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)(k_front_reg<256, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)(k_front_reg<256, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)(k_front_reg<256, 6, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)(k_front_reg<256, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
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
        ready = true; return true;
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
    int grid1d(long long n) const { long long g = (n + 255) / 256; return (int)std::min<long long>(std::max<long long>(g, 1), 2048); }


    // one (level, class) bucket of fronts
    bool launch_bucket(int lv, int fc, int b0, int b1, int top_mode, int mm, int kk, int tiles, int tiles64) {
        const int nb = b1 - b0;
        if (fc != FC_BIG && !drain_chain()) return false;        // (a level with small fronts is never a chain level)
        const size_t rl = reg_lds[(size_t)lv * FC_COUNT + fc];
        if (fc == FC_WAVE) {
            const bool sg = b0 == S->level_ptr[(size_t)lv * FC_COUNT + FC_WAVE];                            // single-GPU schedule: the bucket is sorted by order
            const int nt = sg ? tiny_split[lv] : 0;
            int fl = top_mode;
            if (V.fastpiv && wave_mmin[lv] <= 16) {      // fronts of order <= 16: four per wavefront on the static-order path first; what it accepts is skipped below
                const int n16 = sg ? tiny16[lv] : nb;
                if (n16 > 0) { LAUNCH(KK_FRONT_WAVE, k_front_dpp16, dim3((n16 + 3) / 4), dim3(64), 0, stream, V, b0, n16, top_mode); fl |= 2; }
            }
            if (nt > 0) LAUNCH(KK_FRONT_WAVE, (k_front_reg<64, 2>), dim3(nt), dim3(64), rl, stream, V, b0, fl);
            if (nb - nt > 0) LAUNCH(KK_FRONT_WAVE, (k_front_reg<64, 4>), dim3(nb - nt), dim3(64), rl, stream, V, b0 + nt, fl);
        } else if (fc == FC_LDS64) {
            LAUNCH(KK_FRONT_LDS64, (k_front_reg<64, 8>), dim3(nb), dim3(64), rl, stream, V, b0, top_mode);
        } else if (fc == FC_LDS128) {
            const int nm = (b0 == S->level_ptr[(size_t)lv * FC_COUNT + FC_LDS128]) ? mid_split[lv] : 0;    // single-GPU schedule only
            const int fl = top_mode | (V.fastpiv ? 2 : 0);
            if (V.fastpiv) {      // fronts with <= 16 pivots: static-order path first; what it accepts is skipped by the launch behind it
                if (nm > 0) LAUNCH(KK_FRONT_LDS128, (k_front_reg<256, 6, true>), dim3(nm), dim3(256), mid_lds[lv], stream, V, b0, top_mode);
                if (nb - nm > 0) LAUNCH(KK_FRONT_LDS128, (k_front_reg<256, 8, true>), dim3(nb - nm), dim3(256), rl, stream, V, b0 + nm, top_mode);
            }
            if (nm > 0) LAUNCH(KK_FRONT_LDS128, (k_front_reg<256, 6>), dim3(nm), dim3(256), mid_lds[lv], stream, V, b0, fl);
            if (nb - nm > 0) LAUNCH(KK_FRONT_LDS128, (k_front_reg<256, 8>), dim3(nb - nm), dim3(256), rl, stream, V, b0 + nm, fl);
        } else {
            const bool single = (b0 == S->level_ptr[(size_t)lv * FC_COUNT + FC_BIG]) && !multi;       // the single-GPU schedule
            if ((single || multi) && grouped && lv >= S->grp_cut_level) {
                // the chain groups whose first link sits on this level, one launch each (the multi-GPU schedules have their own group lists)
                if (!drain_chain()) return false;
                if (!(single && lv_asm_skip[lv])) LAUNCH(KK_BIG_ASSEMBLE, k_big_assemble, dim3((mm + 3) / 4, nb), dim3(256), 0, stream, V, b0, top_mode);
                return launch_groups(lv, single ? gs_single : (top_mode ? *gs_cur : gs_local));
            }
            if (!single) { const bool sm = mm <= 640; return launch_big(lv, b0, sm ? b1 : b0, b1, top_mode, mm, kk, sm ? tiles64 : 0, sm ? 0 : tiles, false); }
            return launch_big(lv, b0, b0 + big_split[lv], b1, top_mode, mm, kk, part_tiles[0][lv], part_tiles[1][lv], true);
        }
        return true;
    }
    // the big-front launches of one level: assembly, pivot blocks and TRSM over the whole list [b0, b1); the trailing update with
    // 64 x 64 tiles / 256 threads on the fronts [b0, bs) of order <= 1024 (a handful of tiles, K = 16..64 each) and with
    // 128 x 128 tiles / 1024 threads on [bs, b1)
    bool drain_chain() {
        if (ch_bulk_pending) { HIPCHK(hipStreamWaitEvent(stream, ch_bulk_last, 0)); ch_bulk_pending = false; }
        if (ch_far_pending) { HIPCHK(hipStreamWaitEvent(stream, ch_far_last, 0)); ch_far_pending = false; }
        return true;
    }
    // one level of pure chain links with look-ahead (see the member comment): [b0, bs) fronts of order <= 1024, [bs, b1) larger
    bool launch_big_chain(int lv, int b0, int bs, int b1, int mm, int kk, int tiles_small, int tiles) {
        const int nball = b1 - b0, nrb = (mm + 63) / 64;
        if (la_pending) { HIPCHK(hipStreamWaitEvent(stream, la_last, 0)); la_pending = false; }
        // ---- critical path (main stream) ----
        hipLaunchKernelGGL(k_big_diag_reg<4>, dim3(nball), dim3(256), diag_lds_bytes(kk, 64), stream, V, b0);
        HIPCHK(hipEventRecord(chD[lv], stream));
        if (ch_bulk_pending) HIPCHK(hipStreamWaitEvent(stream, ch_bulk_last, 0));      // this panel's first row block was finalised by the previous level's bulk update
        hipLaunchKernelGGL(k_big_trsm<false>, dim3(1, nball), dim3(256), trsm_lds(kk), stream, V, b0, 0);
        hipLaunchKernelGGL(k_big_schur64, dim3(1, nball), dim3(256), 0, stream, V, b0, 1);
        HIPCHK(hipEventRecord(chLA[lv], stream));
        // ---- bulk (stream3): the rest of the panel solve, then the trailing update minus the block done above ----
        HIPCHK(hipStreamWaitEvent(stream3, chD[lv], 0));
        if (nrb > 1) hipLaunchKernelGGL(k_big_trsm<false>, dim3(nrb - 1, nball), dim3(256), trsm_lds(kk), stream3, V, b0, 1);
        HIPCHK(hipStreamWaitEvent(stream3, chLA[lv], 0));
        if (bs > b0 && tiles_small > 0) hipLaunchKernelGGL(k_big_schur64, dim3(tiles_small, bs - b0), dim3(256), 0, stream3, V, b0, 2);
        if (b1 > bs) {
            const int nb = b1 - bs;
            if (la_full[lv] && ch_far_pending) { HIPCHK(hipStreamWaitEvent(stream3, ch_far_last, 0)); ch_far_pending = false; }   // a full update touches what the previous far part writes
            if (la_tiles2[lv] > 0) {
                hipLaunchKernelGGL(k_big_schur, dim3(la_tiles1[lv], nb), dim3(1024), 0, stream3, V, bs, 1, 0, 1);
                HIPCHK(hipEventRecord(chG1[lv], stream3));
                HIPCHK(hipStreamWaitEvent(stream2, chG1[lv], 0));
                hipLaunchKernelGGL(k_big_schur, dim3(std::min(la_tiles2[lv], la_wgs), nb), dim3(1024), 0, stream2, V, bs, 2, la_tiles2[lv], 0);
                HIPCHK(hipEventRecord(chFar[lv], stream2));
                ch_far_last = chFar[lv]; ch_far_pending = true;
            } else if (tiles > 0) hipLaunchKernelGGL(k_big_schur, dim3(tiles, nb), dim3(1024), 0, stream3, V, bs, 0, 0, 1);
        }
        HIPCHK(hipEventRecord(chN[lv], stream3));
        ch_bulk_last = chN[lv]; ch_bulk_pending = true;
        HIPCHK(hipGetLastError());
        return true;
    }
    // the chain groups whose first link sits on level lv: pivot blocks + leading blocks, rows below, rank-(<= 256) updates
    bool launch_groups(int lv, GrpSched& G) {
        const int b0 = G.g0[lv], b1 = G.g1[lv], bs = b0 + G.split[lv];
        if (b1 == b0) return true;
        {
            const int st = (b1 - b0) * (4 + G.nrb[lv]) <= 256 ? 1 : 0;
            LAUNCH(KK_BIG_DIAG, k_grp_fused, dim3(b1 - b0, 4 + G.nrb[lv]), dim3(256), std::max(diag_lds_bytes(64, 64), GRP_DB_BYTES + trsm_lds_bytes(64, st != 0)), stream, V, b0, st);
        }
        if (bs > b0 && G.tiles64[lv] > 0) LAUNCH(KK_BIG_SCHUR, k_big_schur64, dim3(G.tiles64[lv], bs - b0), dim3(256), 0, stream, V, b0, 0);
        if (b1 == bs) return true;
        const int nb = b1 - bs;
        if (la_pending) { HIPCHK(hipStreamWaitEvent(stream, la_last, 0)); la_pending = false; }      // a full update may touch what an earlier part 2 is still writing
        if (G.la2[lv] > 0 && !prof_on) {
            LAUNCH(KK_BIG_SCHUR, k_big_schur, dim3(G.la1[lv], nb), dim3(1024), 0, stream, V, bs, 1, 0, 0);
            HIPCHK(hipEventRecord(G.evA[lv], stream));
            HIPCHK(hipStreamWaitEvent(stream2, G.evA[lv], 0));
            hipLaunchKernelGGL(k_big_schur, dim3(std::min(G.la2[lv], la_wgs), nb), dim3(1024), 0, stream2, V, bs, 2, G.la2[lv], 0);
            HIPCHK(hipEventRecord(G.evB[lv], stream2));
            la_last = G.evB[lv]; la_pending = true;
        } else if (G.tiles[lv] > 0) LAUNCH(KK_BIG_SCHUR, k_big_schur, dim3(G.tiles[lv], nb), dim3(1024), 0, stream, V, bs, 0, 0, 0);
        return true;
    }
    bool launch_big(int lv, int b0, int bs, int b1, int top_mode, int mm, int kk, int tiles_small, int tiles, bool single) {
        if (single && lv_chain[lv] && !prof_on && !top_mode) return launch_big_chain(lv, b0, bs, b1, mm, kk, tiles_small, tiles);
        if (!drain_chain()) return false;
        const int nball = b1 - b0;
        if (!(single && lv_asm_skip[lv])) LAUNCH(KK_BIG_ASSEMBLE, k_big_assemble, dim3((mm + 3) / 4, nball), dim3(256), 0, stream, V, b0, top_mode);
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
        if (single && la_tiles2[lv] > 0 && !prof_on) {
            LAUNCH(KK_BIG_SCHUR, k_big_schur, dim3(la_tiles1[lv], nb), dim3(1024), 0, stream, V, b0, 1, 0, 0);
            HIPCHK(hipEventRecord(la_evA[lv], stream));
            HIPCHK(hipStreamWaitEvent(stream2, la_evA[lv], 0));
            hipLaunchKernelGGL(k_big_schur, dim3(std::min(la_tiles2[lv], la_wgs), nb), dim3(1024), 0, stream2, V, b0, 2, la_tiles2[lv], 0);
            HIPCHK(hipEventRecord(la_evB[lv], stream2));
            la_last = la_evB[lv]; la_pending = true;
        } else if (tiles > 0) LAUNCH(KK_BIG_SCHUR, k_big_schur, dim3(tiles, nb), dim3(1024), 0, stream, V, b0, 0, 0, 0);
        return true;
    }

    // symmetric scaling of the gathered values (mode 0 none / 1 Ruiz, 4 Jacobi-style sweeps ping-ponging between two buffers /
    // 2 the caller's factors) and the column norms of the scaled matrix that anchor the zero-pivot test
    void enqueue_scaling() {
        const Symbolic& Sy = *S; const int n = Sy.n;
        LAUNCH(KK_GATHER_SCALE, k_abs_rowview, dim3(grid1d(V.rslot_len)), dim3(256), 0, stream, V);
        if (opt.scaling >= 2) LAUNCH(KK_GATHER_SCALE, k_user_scale, dim3(grid1d(n)), dim3(256), 0, stream, V, (const double*)d_user_scale);     // 2: the caller's factors, 3: matching (computed just before)
        else if (opt.scaling) {
            LAUNCH(KK_GATHER_SCALE, k_ruiz_sweep, dim3(grid1d(8ll * n)), dim3(256), 0, stream, V, (const double*)nullptr, V.scale2, (double*)nullptr);
            LAUNCH(KK_GATHER_SCALE, k_ruiz_sweep, dim3(grid1d(8ll * n)), dim3(256), 0, stream, V, (const double*)V.scale2, V.scale, (double*)nullptr);
            LAUNCH(KK_GATHER_SCALE, k_ruiz_sweep, dim3(grid1d(8ll * n)), dim3(256), 0, stream, V, (const double*)V.scale, V.scale2, (double*)nullptr);
            LAUNCH(KK_GATHER_SCALE, k_ruiz_sweep, dim3(grid1d(8ll * n)), dim3(256), 0, stream, V, (const double*)V.scale2, V.scale, V.cnorm);
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
        for (int lv = 0; lv < Sy.num_levels; ++lv) {
            for (int fc = 0; fc < FC_COUNT; ++fc) {
                const int b0 = Sy.level_ptr[(size_t)lv * FC_COUNT + fc], b1 = Sy.level_ptr[(size_t)lv * FC_COUNT + fc + 1];
                if (b1 == b0) continue;
                launch_bucket(lv, fc, b0, b1, 0, big_maxm[lv], big_maxk[lv], big_tiles[lv], big_tiles64[lv]);
            }
        }
        if (la_pending) { HIPCHK(hipStreamWaitEvent(stream, la_last, 0)); la_pending = false; }
        if (!drain_chain()) return false;
        LAUNCH(KK_STATS, k_zero_i32, dim3(1), dim3(64), 0, stream, d_stats, 4);
        LAUNCH(KK_STATS, k_reduce_stats, dim3(std::min(64, (Sy.num_sn + 255) / 256)), dim3(256), 0, stream, V.fstat, V.sn_owner, Sy.num_sn, -2, d_stats);
        HIPCHK(hipGetLastError());
        return true;
    }

    bool factor(const double* dvals, bool reuse, FactorStats& st) {
        DeviceGuard guard(dev);
        if (!ready) { if (err_.empty()) err_ = "factor: solver not set up (no device?)"; return false; }
        if (multi) return factor_dist(dvals, reuse, st);          // needs a communicator (set_comm_*), fails loudly otherwise
        const Symbolic& Sy = *S;
        V.pivtol = opt.pivtol; V.pivtol2 = std::max(opt.pivtol, opt.pivtolmax); V.small = opt.small;
        if (!reuse) {
            if (dvals) HIPCHK(hipMemcpyAsync((void*)V.tvals, dvals, Sy.nnz_in * sizeof(double), hipMemcpyDeviceToDevice, stream));
            else       HIPCHK(hipMemcpyAsync((void*)V.tvals, h_vals, Sy.nnz_in * sizeof(double), hipMemcpyHostToDevice, stream));
            have_values = true;
        } else if (!have_values) { err_ = "refactor: no values on the device yet"; return false; }
        if (opt.scaling == 3 && Sy.n > 0 && !compute_matching_scaling()) return false;
        HIPCHK(hipEventRecord(ev0, stream));
        // A factorisation with look-ahead forks onto the second stream: it is launched eagerly (measured equal to the graph
        // replay on these ~10^3-launch sequences, whose kernels are long), because a two-stream hipGraph replays up to 1.5x
        // slower once another solver's graphs have been created and destroyed in the same process (ROCm 7.2).
        if (opt.use_graph && !la_any) {
            if (!g_factor || graph_pivtol != V.pivtol || graph_pivtol2 != V.pivtol2) {
                if (g_factor) { (void)hipGraphExecDestroy(g_factor); g_factor = nullptr; }
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
        HIPCHK(hipMemcpyAsync(h_stats, d_stats, 8 * sizeof(int), hipMemcpyDeviceToHost, stream));
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
            for (int lv = 0; lv < Sy.num_levels; ++lv) {
                if (seg_at_lv0[lv] >= 0) {      // a run of pure chain levels: one sync-free launch for all of them
                    const ChainSeg& sg = chain_segs[seg_at_lv0[lv]];
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
            for (int lv = Sy.num_levels - 1; lv >= 0; --lv) {
                if (seg_at_lv1[lv] >= 0) {
                    const ChainSeg& sg = chain_segs[seg_at_lv1[lv]];
                    LAUNCH(KK_BWD_BIG, k_bwd_chain, dim3(sg.nwg_b), dim3(320), 0, stream, V, sg.wgb0);
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
        }
        if (nref > 0) LAUNCH(KK_SOLVE_PERM, k_refine_finish, dim3(grid1d(n)), dim3(256), 0, stream, V);
        HIPCHK(hipGetLastError());
        return true;
    }

    bool solve_device(int nrhs, const double* dsrc, int lds_, double* drhs, int ld, bool timed) {
        DeviceGuard guard(dev);
        if (!ready) { if (err_.empty()) err_ = "solve: solver not set up"; return false; }
        if (multi) return solve_dist(nrhs, dsrc, lds_, drhs, ld);
        if (timed) HIPCHK(hipEventRecord(ev0, stream));
        for (int r = 0; r < nrhs; ++r) {
            double* col = drhs + (size_t)r * ld;
            const double* src = dsrc + (size_t)r * lds_;
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
                    if (timed && r == 0) HIPCHK(hipEventRecord(ev0, stream));
                }
                hipLaunchKernelGGL(k_load_rhs, dim3(grid1d(S->n)), dim3(256), 0, stream, V, src);
                HIPCHK(hipGraphLaunch(g_solve, stream));
                hipLaunchKernelGGL(k_store_sol, dim3(grid1d(S->n)), dim3(256), 0, stream, V, col);
            } else if (!enqueue_solve(src, col)) return false;
        }
        if (timed) {
            HIPCHK(hipEventRecord(ev1, stream));
            if (!chain_segs.empty()) HIPCHK(hipMemcpyAsync(h_stats + 6, V.sepoch + 1, sizeof(int), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ev0, ev1)); solve_ms = ms;
            if (V.strace) {
                std::vector<unsigned long long> h(strace_n);
                HIPCHK(hipMemcpy(h.data(), V.strace, strace_n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                if (FILE* f = fopen(getenv("MI355X_KKT_SOLVE_TRACE"), "w")) {
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
        if (opt.scaling == 3 && n > 0 && !compute_matching_scaling()) return false;
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
        hipLaunchKernelGGL(k_reduce_stats, dim3(1), dim3(256), 0, stream, V.fstat, V.sn_owner, S->num_sn, opt.rank, d_stats);
        hipLaunchKernelGGL(k_reduce_stats, dim3(1), dim3(256), 0, stream, V.fstat, V.sn_owner, S->num_sn, -1, d_stats);
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
            if (aend[d] > abeg[d] && !allreduce(V.arena + abeg[d], aend[d] - abeg[d], 0)) return false;
            if (!enqueue_factor_step(d)) return false;
        }
        if (!enqueue_stats()) return false;
        if (!allreduce(d_stats, 8, 1)) return false;
        HIPCHK(hipEventRecord(ev1, stream));
        HIPCHK(hipMemcpyAsync(h_stats, d_stats, 8 * sizeof(int), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ev0, ev1)); factor_ms = ms;
        st.num_neg = h_stats[0]; st.num_zero = h_stats[1]; st.num_two = h_stats[2]; st.num_small = h_stats[3]; st.u_sensitive = h_stats[4] != 0; st.num_fast = h_stats[7];
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
                if (tend[d] > tbeg[d] && !allreduce(V.top_rhs + tbeg[d], tend[d] - tbeg[d], 0)) return false;
                if (!launch_solve_sweep(sch_stage[d], true, 1, d == 0)) return false;
                if (d > 0 && !report_to_top_rhs(1 + d)) return false;
            }
            for (int d = 0; d < ndepth; ++d) if (!launch_solve_sweep(sch_stage[d], false, 1, d == 0)) return false;
            if (!launch_solve_sweep(sch_local, false, 0)) return false;
            hipLaunchKernelGGL(k_store_sol_mg, dim3(grid1d(S->n)), dim3(256), 0, stream, V, col);
            HIPCHK(hipGetLastError());
            if (!allreduce(col, S->n, 0)) return false;
        }
        HIPCHK(hipEventRecord(ev1, stream)); HIPCHK(hipStreamSynchronize(stream));
        float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ev0, ev1)); solve_ms = ms;
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
        for (int i = 0; i < S->n; ++i) if (z[i]) out.push_back(S->perm[i]);
        std::sort(out.begin(), out.end());
        return true;
    }
    bool debug_clocks(unsigned long long* out) {
        DeviceGuard guard(dev);
        if (!V.dbg) { err_ = "debug clocks not enabled (MI355X_KKT_DEBUG_CLOCKS=1)"; return false; }
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
        if (d_rhs_cap < n) { if (d_rhs) (void)hipFree(d_rhs); d_rhs = nullptr; HIPCHK(hipMalloc((void**)&d_rhs, std::max<size_t>(n, 1) * sizeof(double))); d_rhs_cap = n;
                             if (g_solve) { (void)hipGraphExecDestroy(g_solve); g_solve = nullptr; } }
        double total = 0;
        for (int r = 0; r < nrhs; ++r) {
            HIPCHK(hipMemcpyAsync(d_rhs, rhs + (size_t)r * ld, n * sizeof(double), hipMemcpyHostToDevice, stream));
            if (!solve_device(1, d_rhs, (int)n, d_rhs, (int)n, true)) return false;
            total += solve_ms;
            HIPCHK(hipMemcpyAsync(rhs + (size_t)r * ld, d_rhs, n * sizeof(double), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
        }
        solve_ms = total;
        return true;
    }
};

Numeric::Numeric() : p_(new NumericImpl) {}
Numeric::~Numeric() { delete p_; }
bool Numeric::setup(const Symbolic& S, const NumericOptions& opt) { return p_->setup(S, opt); }
double* Numeric::values_buffer() { return p_->h_vals; }
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
void Numeric::set_pivtol(double u) { p_->opt.pivtol = u; }
void Numeric::set_pivtolmax(double u) { p_->opt.pivtolmax = u; }
double Numeric::last_factor_ms() const { return p_->factor_ms; }
double Numeric::last_solve_ms() const { return p_->solve_ms; }
const std::string& Numeric::error() const { return p_->err_; }
bool Numeric::profile(int reps, double* ms, int* launches) { return p_->profile(reps, ms, launches); }
bool Numeric::debug_clocks(unsigned long long* out) { return p_->debug_clocks(out); }
bool Numeric::factor_local(const double* dvals) { return p_->factor_local(dvals); }
bool Numeric::top_arena(double** d, int64_t* nd) { if (!p_->multi) { p_->err_ = "top_arena: not a multi-GPU handle"; return false; } *d = p_->V.arena; *nd = p_->arena_doubles; return true; }
bool Numeric::factor_top(FactorStats& st) { return p_->factor_top(st); }
bool Numeric::solve_fwd_local(double* drhs) { return p_->solve_fwd_local(drhs); }
bool Numeric::top_rhs(double** d, int64_t* nd) { if (!p_->multi) { p_->err_ = "top_rhs: not a multi-GPU handle"; return false; } *d = p_->V.top_rhs; *nd = p_->toprhs_doubles; return true; }
bool Numeric::solve_top_and_bwd(double* drhs) { return p_->solve_top_and_bwd(drhs); }
bool Numeric::set_scaling(int mode, const double* user) { return p_->set_scaling(mode, user); }
bool Numeric::get_scaling(double* out) { return p_->get_scaling(out); }
bool Numeric::zero_pivots(std::vector<int>& out) { return p_->zero_pivots(out); }
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

// symbolic.h -- host-side symbolic analysis for the MI355X multifrontal LDL^T.
//
// Replaces, for our backend, what the reference delegates to its third-party solvers in
// InitializeStructure/SymbolicFactorization (IpMa27TSolverInterface.cpp:358-470 -> ma27ad,
// IpMumpsSolverInterface.cpp:385-446 -> MUMPS job 1, IpMa97SolverInterface.cpp:567 -> ma97_analyse)
// and what TripletToCSRConverter::InitializeConverter does on the host
// (IpTripletToCSRConverter.cpp:46-335): duplicate merging and triangle canonicalisation.
//
// Everything here is one-time per sparsity structure: the output is a set of flat
// int32/int64 arrays that are uploaded once and drive the device kernels.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>
#include <string>

namespace mi355x {

// ---------------------------------------------------------------------------------------------------------------------------------
// Memory of the analysis.  The analysis builds and drops a few dozen arrays of the size of the matrix (10^7 .. 10^8 bytes each); from
// the C library every one of them is a fresh anonymous mapping whose pages are faulted in -- by many threads at once, serialised in the
// kernel -- and handed back on free: a third of the analysis time at n = 10^6 (measured: 3.9 s -> 2.6 s on 8 cores with glibc told to
// keep the memory, MALLOC_MMAP_THRESHOLD_ / MALLOC_TRIM_THRESHOLD_).  A library must not retune the process's allocator, so the big
// blocks of the analysis are recycled here instead: while an analysis runs, a freed block of >= 256 KiB is kept and handed to the next
// request it fits; when the analysis returns, what is still cached goes back to the C library.  Blocks are plain malloc blocks at all
// times (a vector that outlives the analysis -- the Symbolic arrays -- is freed normally).
// ---------------------------------------------------------------------------------------------------------------------------------
struct BlockCache {
    static void* get(size_t bytes);
    static void  put(void* p, size_t bytes) noexcept;
    struct Scope { Scope(); ~Scope(); };          // recycling is on while at least one Scope lives (one per running analysis)
};
template <class T> struct CacheAlloc {
    using value_type = T;
    CacheAlloc() = default;
    template <class U> CacheAlloc(const CacheAlloc<U>&) noexcept {}
    T* allocate(size_t n) { return static_cast<T*>(BlockCache::get(n * sizeof(T))); }
    void deallocate(T* p, size_t n) noexcept { BlockCache::put(p, n * sizeof(T)); }
    template <class U> bool operator==(const CacheAlloc<U>&) const noexcept { return true; }
    template <class U> bool operator!=(const CacheAlloc<U>&) const noexcept { return false; }
};
template <class T> using avec = std::vector<T, CacheAlloc<T>>;

struct SymbolicOptions {
    int    index_base  = 1;
    int    ordering    = 0;    // 0 ND+MD, 1 MD, 2 natural
    int    matching    = 1;
    int    nd_leaf     = 32;
    int    nemin       = 8;
    int    max_sn_cols = 64;
    int    leaf_cols   = 0;    // whole elimination subtrees with at most this many columns become ONE supernode
    int    tree_merge  = 0;    // merge small non-contiguous child supernodes into the parent: 1 on, 0 off (default), -1 auto (n <= 4e5)
    int    solve_group = 0;    // 1: the triangular solves also work per chain group (default: per link, measured faster)
    int    chain_group = 4;    // links of an in-place separator chain handled as one unit (1 = off, max 4)
    int    chain_purify = 1;   // the small side children of in-place chain links are assembled into the chain's first link instead (symbolic.cpp 9b): pure chains => chain groups
    int    wide_panels = 0;    // 1: separator fronts of order >= 512 get 128-column panels (kernels support it; default off)
    int    nranks      = 1;
    int    subcube     = 0;    // multi-GPU: 1 = subtree-to-subcube mapping (a top front is replicated on the ranks beneath it only), 0 = one replicated top
    int    verbose     = 0;
};

// front size classes (kernel selection); see numeric.hip
enum FrontClass : int { FC_WAVE = 0,   // m <= 32  : one wavefront, front in LDS
                        FC_LDS64 = 1,  // m <= 64  : 256 threads, front in LDS (32 KiB)
                        FC_LDS128 = 2, // m <= 128 : 256 threads, front in LDS (128 KiB)
                        FC_BIG = 3,    // m  > 128 : blocked global-memory path (MFMA trailing updates)
                        FC_COUNT = 4 };

struct Symbolic {
    int n = 0, nnz_in = 0, nnz_a = 0;
    // permutation: perm[new] = old, iperm[old] = new
    avec<int> perm, iperm;
    avec<int> pair_of;              // old index of 2x2 partner or -1
    int num_pairs = 0;
    // permuted lower CSC pattern (row >= col, sorted rows, diagonal always present & first)
    avec<int> acolptr, arow;        // [n+1], [nnz_a]
    avec<int> trip2slot;            // [nnz_in] triplet -> slot in arow/aval
    // duplicate lists grouped by slot (device gather-sum in fixed order => deterministic)
    avec<int> dup_ptr, dup_src;     // [nnz_a+1], [nnz_in]
    // symmetric index pairs for scaling: slot -> (row,col) are arow / column of slot
    avec<int> acol;                 // [nnz_a] column of each slot (permuted numbering)
    // full symmetric row view of the pattern: for every row i the slots (q) that carry an entry of row i, i.e. column i's
    // own slots and the slots (i, c < i) of earlier columns; lets the equilibration run as a gather (no atomics)
    avec<int> rslot_ptr, rslot_idx; // [n+1], [2*nnz_a - n]  symmetric row view: slots of the entries of row i (both triangles)
    avec<int> rslot_col;            // [2*nnz_a - n] the other index of each of them
    // supernodes
    int num_sn = 0;
    avec<int> sn_colptr;            // [num_sn+1] pivot column ranges (permuted numbering)
    avec<int> sn_of;                // [n] supernode of permuted column
    avec<int> sn_rowptr;            // [num_sn+1] into sn_rows
    avec<int> sn_rows;              // front row lists: k pivots then sorted update rows (permuted numbering)
    avec<int> rel;                  // aligned with sn_rows: local row in PARENT front, -1 for pivot rows
    avec<int> sn_parent;            // [num_sn] or -1
    avec<int> child_ptr, child_idx; // children lists
    avec<int> sn_level;             // height-based level (leaves = 0)
    avec<int> sn_class;             // FrontClass
    avec<int64_t> panel_off;        // [num_sn] offset (doubles) of the m x k panel in L storage (ld = m)
    avec<int64_t> cb_off;           // [num_sn] offset (doubles) of the (m-k)^2 contribution block (ld = sn_ldt)
    // In-place separator chains: a BIG front whose row set equals the update rows of a BIG child is not re-assembled -- it
    // LIVES in that child's contribution block (panel = its first k columns, own contribution block = the trailing part),
    // i.e. a right-looking blocked LDL^T of the separator front.  panel_off of such a front points into the cb pool
    // (offset >= l_doubles; L and cb are one allocation) and both leading dimensions are inherited from the child.
    avec<int> sn_ldp, sn_ldt;       // leading dimensions of the panel / of the contribution block
    avec<int> alias_child;          // the child whose contribution block this front lives in, or -1
    avec<int> grp_pos, grp_rem;     // chain groups: position of the front in its group; columns of the LATER links (0 = last link)
    avec<int64_t> cv_off;           // [num_sn] offset of the front's forward-solve vector (in-place chains share one)
    avec<int64_t> gpart_off;        // [num_sn] (group-last BIG fronts) offset of the backward partial sums
    int64_t cvec_doubles = 0, gpart_doubles = 0;
    int solve_group = 0;                   // copy of the option: solves per chain group (1) or per link (0)
    int grp_cut_level = 0;                 // from this tree level up every level has only a handful of BIG fronts (the latency-bound top of the tree):
                                           // chain groups do not straddle it, the numeric phase factors the groups above it in one launch each
    avec<int64_t> wb_off;           // [num_sn] offset (doubles) of the m x k scaled-panel copy W = L*D of a BIG front
                                           // inside the per-level scratch (reused level after level), -1 otherwise
    int64_t wbuf_doubles = 0;
    avec<int64_t> minv_off;         // [num_sn] offset (doubles) of the k x k inverse of the unit-lower pivot block
    int64_t minv_doubles = 0;
    avec<int> apos;                 // [nnz_a] local position (row + col*m) of each slot inside its panel
    // level schedule: fronts sorted by (level, class)
    int num_levels = 0;
    avec<int> level_ptr;            // [num_levels*FC_COUNT + 1] into level_sn: bucket (level, class)
    avec<int> level_sn;             // [num_sn]
    // multi-GPU ownership: rank owning the supernode (subtree sharding) or -1 = replicated top
    avec<int> sn_owner;
    // ... and the range of ranks [sn_glo, sn_glo + sn_gsz) that holds the front (one rank for an owned front, all ranks for the classic replicated
    // top, the ranks beneath it with the subtree-to-subcube mapping); sn_gdepth: bisections above that range = exchange step the front belongs to
    avec<int> sn_glo, sn_gsz, sn_gdepth;
    int num_gdepths = 1;
    // statistics
    int64_t nnz_l = 0, flops_factor = 0, sum_sn_rows = 0, cb_doubles = 0, l_doubles = 0;
    int     cb_window = 0;                 // > 0: contribution blocks are RECYCLED over the level schedule (symbolic.cpp step 12a); the numeric schedule joins its side streams every cb_window levels
    int64_t cb_resident_doubles = 0, cb_plain_doubles = 0;      // the part of cb_doubles that is never reused (chain hosts, small fronts); what cb_doubles would be with every block resident
    int num_rehung = 0;        // side children of chain links assembled further down their chain (SymbolicOptions::chain_purify)
    int maxfront = 0, maxsupernode = 0, num_big = 0;
    double time_analyse = 0;
    std::string error;
};

// Build everything.  row/col in the caller's numbering (index_base), triplet (either triangle,
// duplicates allowed) or CSR-upper (row = ia[n+1], col = ja[nnz]).  vals may be null.
bool analyse(Symbolic& S, const SymbolicOptions& opt, int n, int nnz, const int* row, const int* col,
             int format, const double* vals);

// Delayed pivoting across fronts as an edit of the supernode partition (symbolic.cpp): the columns `marked` (CURRENT permuted
// numbering of `cur`) that a factorisation could not pivot leave their supernode and join the parent's; `out` is the complete
// analysis of the edited structure.  *moved = columns actually moved (a root front has no parent).  false + out.error when nothing
// moved or on an internal error.
// hops[q] (optional, parallel to marked): tree levels column marked[q] climbs (1 = to the parent; a column that has failed before climbs further).
bool restructure_delays(const Symbolic& cur, const SymbolicOptions& opt, const std::vector<int>& marked, const std::vector<int>& hops, Symbolic& out, int* moved,
                        std::vector<char>* acted = nullptr);      // acted[q]: marked[q] did move

} // namespace mi355x

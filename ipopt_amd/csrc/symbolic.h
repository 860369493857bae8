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
#include <cstdint>
#include <vector>
#include <string>

namespace mi355x {

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
    std::vector<int> perm, iperm;
    std::vector<int> pair_of;              // old index of 2x2 partner or -1
    int num_pairs = 0;
    // permuted lower CSC pattern (row >= col, sorted rows, diagonal always present & first)
    std::vector<int> acolptr, arow;        // [n+1], [nnz_a]
    std::vector<int> trip2slot;            // [nnz_in] triplet -> slot in arow/aval
    // duplicate lists grouped by slot (device gather-sum in fixed order => deterministic)
    std::vector<int> dup_ptr, dup_src;     // [nnz_a+1], [nnz_in]
    // symmetric index pairs for scaling: slot -> (row,col) are arow / column of slot
    std::vector<int> acol;                 // [nnz_a] column of each slot (permuted numbering)
    // full symmetric row view of the pattern: for every row i the slots (q) that carry an entry of row i, i.e. column i's
    // own slots and the slots (i, c < i) of earlier columns; lets the equilibration run as a gather (no atomics)
    std::vector<int> rslot_ptr, rslot_idx; // [n+1], [2*nnz_a - n]  symmetric row view: slots of the entries of row i (both triangles)
    std::vector<int> rslot_col;            // [2*nnz_a - n] the other index of each of them
    // supernodes
    int num_sn = 0;
    std::vector<int> sn_colptr;            // [num_sn+1] pivot column ranges (permuted numbering)
    std::vector<int> sn_of;                // [n] supernode of permuted column
    std::vector<int> sn_rowptr;            // [num_sn+1] into sn_rows
    std::vector<int> sn_rows;              // front row lists: k pivots then sorted update rows (permuted numbering)
    std::vector<int> rel;                  // aligned with sn_rows: local row in PARENT front, -1 for pivot rows
    std::vector<int> sn_parent;            // [num_sn] or -1
    std::vector<int> child_ptr, child_idx; // children lists
    std::vector<int> sn_level;             // height-based level (leaves = 0)
    std::vector<int> sn_class;             // FrontClass
    std::vector<int64_t> panel_off;        // [num_sn] offset (doubles) of the m x k panel in L storage (ld = m)
    std::vector<int64_t> cb_off;           // [num_sn] offset (doubles) of the (m-k)^2 contribution block (ld = sn_ldt)
    // In-place separator chains: a BIG front whose row set equals the update rows of a BIG child is not re-assembled -- it
    // LIVES in that child's contribution block (panel = its first k columns, own contribution block = the trailing part),
    // i.e. a right-looking blocked LDL^T of the separator front.  panel_off of such a front points into the cb pool
    // (offset >= l_doubles; L and cb are one allocation) and both leading dimensions are inherited from the child.
    std::vector<int> sn_ldp, sn_ldt;       // leading dimensions of the panel / of the contribution block
    std::vector<int> alias_child;          // the child whose contribution block this front lives in, or -1
    std::vector<int> grp_pos, grp_rem;     // chain groups: position of the front in its group; columns of the LATER links (0 = last link)
    std::vector<int64_t> cv_off;           // [num_sn] offset of the front's forward-solve vector (in-place chains share one)
    std::vector<int64_t> gpart_off;        // [num_sn] (group-last BIG fronts) offset of the backward partial sums
    int64_t cvec_doubles = 0, gpart_doubles = 0;
    int solve_group = 0;                   // copy of the option: solves per chain group (1) or per link (0)
    int grp_cut_level = 0;                 // from this tree level up every level has only a handful of BIG fronts (the latency-bound top of the tree):
                                           // chain groups do not straddle it, the numeric phase factors the groups above it in one launch each
    std::vector<int64_t> wb_off;           // [num_sn] offset (doubles) of the m x k scaled-panel copy W = L*D of a BIG front
                                           // inside the per-level scratch (reused level after level), -1 otherwise
    int64_t wbuf_doubles = 0;
    std::vector<int64_t> minv_off;         // [num_sn] offset (doubles) of the k x k inverse of the unit-lower pivot block
    int64_t minv_doubles = 0;
    std::vector<int> apos;                 // [nnz_a] local position (row + col*m) of each slot inside its panel
    // level schedule: fronts sorted by (level, class)
    int num_levels = 0;
    std::vector<int> level_ptr;            // [num_levels*FC_COUNT + 1] into level_sn: bucket (level, class)
    std::vector<int> level_sn;             // [num_sn]
    // multi-GPU ownership: rank owning the supernode (subtree sharding) or -1 = replicated top
    std::vector<int> sn_owner;
    // ... and the range of ranks [sn_glo, sn_glo + sn_gsz) that holds the front (one rank for an owned front, all ranks for the classic replicated
    // top, the ranks beneath it with the subtree-to-subcube mapping); sn_gdepth: bisections above that range = exchange step the front belongs to
    std::vector<int> sn_glo, sn_gsz, sn_gdepth;
    int num_gdepths = 1;
    // statistics
    int64_t nnz_l = 0, flops_factor = 0, sum_sn_rows = 0, cb_doubles = 0, l_doubles = 0;
    int maxfront = 0, maxsupernode = 0, num_big = 0;
    double time_analyse = 0;
    std::string error;
};

// Build everything.  row/col in the caller's numbering (index_base), triplet (either triangle,
// duplicates allowed) or CSR-upper (row = ia[n+1], col = ja[nnz]).  vals may be null.
bool analyse(Symbolic& S, const SymbolicOptions& opt, int n, int nnz, const int* row, const int* col,
             int format, const double* vals);

} // namespace mi355x

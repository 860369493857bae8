/*
 * mi355x_kkt.h -- C ABI of the MI355X-native sparse symmetric-indefinite KKT solver
 * (supernodal multifrontal LDL^T + triangular solves, hand-written HIP for gfx950).
 *
 * This is the drop-in boundary for Ipopt's linear-solver plug-in point.  Every entry
 * point replaces one step of the reference's third-party-solver adapters:
 *
 *   mi355x_kkt_create / _destroy   <->  adapter ctor/dtor, e.g. Ma97SolverInterface
 *                                       (reference src/Algorithm/LinearSolvers/IpMa97SolverInterface.cpp:277-301)
 *   mi355x_kkt_analyse             <->  SparseSymLinearSolverInterface::InitializeStructure
 *                                       (IpSparseSymLinearSolverInterface.hpp:139) -> ma97_analyse (IpMa97SolverInterface.cpp:567,674),
 *                                       MUMPS job=1 (IpMumpsSolverInterface.cpp:385-446)
 *   mi355x_kkt_values_buffer       <->  GetValuesArrayPtr (IpSparseSymLinearSolverInterface.hpp:155)
 *   mi355x_kkt_factor              <->  the "new_matrix" half of MultiSolve (hpp:190): ma97_factor (IpMa97SolverInterface.cpp:707),
 *                                       MUMPS job=2 (IpMumpsSolverInterface.cpp:448-541); returns inertia like INFOG(12)/info.num_neg
 *   mi355x_kkt_solve               <->  the back-solve half of MultiSolve: ma97_solve (IpMa97SolverInterface.cpp:790,805),
 *                                       MUMPS job=3 (IpMumpsSolverInterface.cpp:543-583)
 *   mi355x_kkt_increase_quality    <->  IncreaseQuality (hpp:220; u <- u^0.75, IpMa97SolverInterface.cpp:822-854)
 *   mi355x_kkt_set_pivtol          <->  the ma97_u / ma27_pivtol option (IpMa97SolverInterface.cpp:93-106)
 *   mi355x_kkt_get_info            <->  struct ma97_info (hsl_ma97d.h:96-121) / MUMPS INFOG
 *
 * Conventions: plain pointers and sizes only, no C++ types, no exceptions cross this
 * boundary, no global mutable state (re-entrant per handle).  Return value of every
 * int function is a MI355X_KKT_* status that mirrors Ipopt's ESymSolverStatus
 * (IpSymLinearSolver.hpp:19-33): 0 success, 1 singular, 2 wrong inertia (never produced
 * here: the caller compares num_neg), 4 fatal.  All floating point is fp64, all
 * indices int32 (Ipopt's Index), 64-bit counters where nnz(L)/flops can overflow.
 *
 * There is NO CPU fallback: factor/solve fail with MI355X_KKT_FATAL (and a message in
 * mi355x_kkt_last_error) when no gfx950 device is usable.  Only _analyse (symbolic,
 * host C++) and the query functions work without a GPU.
 */
#ifndef MI355X_KKT_H
#define MI355X_KKT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355X_KKT_SUCCESS        0
#define MI355X_KKT_SINGULAR       1
#define MI355X_KKT_WRONG_INERTIA  2
#define MI355X_KKT_CALL_AGAIN     3
#define MI355X_KKT_FATAL          4

/* input formats, cf. EMatrixFormat (IpSparseSymLinearSolverInterface.hpp:102-114) */
#define MI355X_KKT_FMT_TRIPLET    0   /* (row[k],col[k]) k<nnz; either triangle; duplicates are summed */
#define MI355X_KKT_FMT_CSR_UPPER  1   /* row = ia[n+1], col = ja[nnz]; upper-triangular CSR == lower CSC */

typedef struct mi355x_kkt_handle_s* mi355x_kkt_handle;

typedef struct mi355x_kkt_options {
    int    device;          /* HIP device ordinal; -1 = current device                                  */
    int    index_base;      /* 1 (Ipopt/Fortran numbering, default) or 0                                */
    int    ordering;        /* 0 = nested dissection + minimum-degree leaves (default), 1 = MD only,    */
                            /* 2 = natural (identity)                                                   */
    int    matching;        /* 1 (default) = pre-pair zero-diagonal rows with a partner column so that  */
                            /* the pair is one 2x2-capable supernode; 0 = off                           */
    int    scaling;         /* 0 none, 1 (default) = symmetric Ruiz inf-norm equilibration on device,   */
                            /* 3 = maximum-product matching scaling (MC64-style; host, per factorisation), */
                            /* 4 = the same computed at the first factorisation and REUSED until          */
                            /*     IncreaseQuality asks for better (MA97's "...-reuse" switches)          */
                            /* 5 = maximum-product matching scaling ON THE DEVICE (Jacobi auction,        */
                            /*     kernels_match.hip.inc), per factorisation; 6 = the same, reused like 4 */
    int    nd_leaf;         /* ND stops splitting below this many (compressed) nodes; default 32        */
    int    nemin;           /* relaxed-supernode amalgamation: always merge below this #cols; default 8 */
    int    max_sn_cols;     /* cap on columns of an amalgamated supernode; default (and max) 64                */
    double pivtol;          /* relative pivot threshold u (default 1e-8, cf. ma97_u)                    */
    double pivtolmax;       /* upper bound for set_pivtol escalation (default 1e-4)                     */
    double small;           /* |pivot| below this (after scaling) counts as zero (default 1e-20)        */
    int    refine_steps;    /* internal iterative-refinement steps per solve, fp64 residual on device    */
                            /* (default 0: Ipopt runs its own loop, IpPDFullSpaceSolver.cpp:256-346)     */
    int    use_graph;       /* 1 (default) = replay factor/solve launch sequences as hipGraphs          */
    int    nranks;          /* multi-GPU: number of ranks sharing one matrix (default 1)                */
    int    rank;            /* multi-GPU: this rank                                                     */
    int    verbose;         /* 0 silent                                                                 */
    int    leaf_cols;       /* whole elimination subtrees of <= this many columns become one supernode  */
                            /* (default 0 = off; measured: not a win)                                               */
    int    tree_merge;      /* tree amalgamation of small non-contiguous supernodes: 1 on, 0 off,        */
                            /* -1 = on when n <= 400 000.  Default 0: measured NOT to pay    */
    int    wide_panels;     /* 1: 128-column panels on separator fronts of order >= 512 (default 0)      */
    int    chain_group;     /* links of an in-place separator chain per update/solve unit, 1..4 (default 4) */
    int    solve_group;     /* 1: triangular solves per chain group instead of per link (default 0)       */
    int    subcube;         /* multi-GPU: 1 = subtree-to-subcube mapping -- a front of the top of the tree is replicated only */
                            /* on the ranks whose subtrees lie beneath it; 0 (default) = one top replicated on all ranks      */
    int    delay_rounds;    /* delayed pivoting across fronts: columns that fail the threshold test in their front are moved to the */
                            /* parent front's supernode and the matrix is refactored, at most this many times per factor() call   */
                            /* (default 8; 0 = static pivoting: failed pivots are forced and counted in num_small)                */
    int    smart_quality;   /* IncreaseQuality: 0 (default) = raise u whenever it is below pivtolmax, as the reference adapters do     */
                            /* (IpMa97SolverInterface.cpp:822-854); 1 = answer "cannot improve" at once when the last factorisation  */
                            /* recorded that no pivot decision depends on u up to pivtolmax (skips an identical refactorisation)      */
} mi355x_kkt_options;

typedef struct mi355x_kkt_info {
    int     n;
    int     nnz_in;          /* entries handed to analyse                                               */
    int     nnz_a;           /* distinct entries of one triangle (incl. diagonal)                       */
    int64_t nnz_l;           /* entries in L incl. diagonal (sum over supernodes of trapezoid)          */
    int64_t flops_factor;    /* sum_j (c_j-1)(c_j+2), c_j = column count of L             */
    int64_t flops_solve;     /* 4 nnz(L) - 3 n per right-hand side                                      */
    int64_t bytes_factor;    /* 12 nnz(A) + 8 nnz(L) + 4 sum(rows of supernodes)                        */
    int64_t bytes_solve;     /* 2 (8 nnz(L) + 4 sum rows) + 24 n per right-hand side                    */
    int64_t sum_sn_rows;     /* sum over supernodes of front order                                      */
    int64_t cb_doubles;      /* doubles in the contribution-block arena                                 */
    int     num_sn;          /* supernodes                                                              */
    int     num_levels;      /* height of the assembly tree (launch waves per factorisation)            */
    int     maxfront;        /* largest front order                                                     */
    int     maxsupernode;    /* largest number of columns in a supernode                                */
    int     num_pairs;       /* 2x2 pre-pairs from the matching                                         */
    int     num_neg;         /* negative eigenvalues of the last factorisation                          */
    int     num_zero;        /* zero pivots (=> singular) of the last factorisation                     */
    int     num_two;         /* 2x2 pivots of the last factorisation                                    */
    int     num_small;       /* pivots of the last factorisation that FAILED the threshold tests with u */
                             /* and were eliminated anyway (static pivoting): what the delayed-pivot    */
                             /* rounds could not move (root fronts, round / growth limits, or           */
                             /* delay_rounds = 0).  0 = every pivot passed the threshold test           */
    int     num_big_fronts;  /* fronts handled by the blocked (global-memory, MFMA) path                */
    double  time_analyse;    /* host seconds, last analyse                                              */
    double  time_factor_ms;  /* device ms (hip events on the solver's stream), last factor              */
    double  time_solve_ms;   /* device ms, last solve                                                   */
    double  pivtol;          /* the u the next factorisation will use                                   */
    int     u_sensitive;     /* last factorisation: 1 if some pivot decision would differ at u=pivtolmax */
    int     num_fast_blocks; /* last factorisation: pivot blocks of big fronts accepted on the blocked a-posteriori path  */
                             /* (natural order, every multiplier <= 1/max(u, pivtolmax, 1e-4)); the other big fronts'    */
                             /* pivot blocks took the strict threshold-pivoting loop                                     */
    int     num_delayed;     /* columns moved to a parent front since analyse() because they failed the threshold tests in their own (a column */
                             /* that moved up twice counts twice): info.num_delay of MA97 (hsl_ma97d.h:103, IpMa97SolverInterface.cpp:719-779) */
    int     num_restructures;/* structure edits (+ refactorisations) those delays have cost since analyse()                                  */
    double  matching_ms;     /* scaling modes 5 / 6: device time of the last matching-scaling computation (hip events)                                */
    int     matching_rounds; /* ... its auction rounds (all phases)                                                                                    */
    int     matching_unmatched; /* ... columns left without a row when it ended (modes 3 / 4: structural deficiency found by the host algorithm)       */
    double  reserved[3];
} mi355x_kkt_info;

/* fill opts with the defaults documented above */
void mi355x_kkt_default_options(mi355x_kkt_options* opts);

/* create a solver instance; opts may be NULL (defaults) */
int  mi355x_kkt_create(mi355x_kkt_handle* h, const mi355x_kkt_options* opts);
void mi355x_kkt_destroy(mi355x_kkt_handle h);

/* Symbolic analysis, once per sparsity structure.  `vals` (nnz doubles in the same order as
 * row/col, or NULL) is used only to detect zero diagonals for the 2x2 pre-pairing; the
 * Ipopt adapter therefore analyses lazily at the first MultiSolve, as the reference's
 * MUMPS adapter does (IpMumpsSolverInterface.cpp:349-383).  row/col are copied. */
int  mi355x_kkt_analyse(mi355x_kkt_handle h, int n, int nnz, const int* row, const int* col,
                        int format, const double* vals);

/* Pinned host staging buffer of nnz doubles the caller fills before every factor()
 * (GetValuesArrayPtr contract, IpSparseSymLinearSolverInterface.hpp:155). */
double* mi355x_kkt_values_buffer(mi355x_kkt_handle h);

/* Numeric factorisation of the values currently in values_buffer (or, if dvals != NULL, of
 * nnz doubles already resident in device memory).  Outputs may be NULL. */
int  mi355x_kkt_factor(mi355x_kkt_handle h, const double* dvals, int* num_neg, int* num_zero);

/* Re-factor the device-resident copy of the last values (after set_pivtol); cf. MA97/SPRAL
 * adapters that refactor from their private val_ copy (IpMa97SolverInterface.cpp:623). */
int  mi355x_kkt_refactor(mi355x_kkt_handle h, int* num_neg, int* num_zero);

/* ---- device-side value assembly: the next row of the hot path --------------------------------------
 * Replaces, for a host that keeps the pieces of the KKT matrix apart (Ipopt's AugSystemSolver contract,
 * IpAugSystemSolver.hpp:40-120), TripletHelper::FillValues over the whole CompoundSymMatrix (IpTripletHelper.cpp:249-362,
 * driven by IpTSymLinearSolver.cpp:453-533) and the per-factorisation 8 nnz-byte upload: the triplet value array is declared
 * once as a sequence of SEGMENTS that tile [0, nnz) -- for Ipopt: W | D_x | D_s | J_c | D_c | J_d | -I | D_d in the order of
 * IpStdAugSystemSolver.cpp:263-298 -- each with a device-resident source that is uploaded only when ITS content changed
 * (_assembly_buffer: pinned staging, _assembly_upload: async copy).  _factor_assembled forms
 *        values[offset_s + i] = scale_s * source_s[i] + shift_s          (scale_s = 0: the source is not read)
 * on the device (W_factor; delta_x, delta_s, -delta_c, -delta_d as shifts; the (4,2) block as scale 0, shift -1) and factors.
 * An inertia-correction retry (IpPDFullSpaceSolver.cpp:486-640: same matrices, new deltas) therefore uploads NOTHING. */
int  mi355x_kkt_assembly_define(mi355x_kkt_handle h, int nseg, const int64_t* offset, const int64_t* length);   /* nseg <= 16 */
double* mi355x_kkt_assembly_buffer(mi355x_kkt_handle h, int seg);
int  mi355x_kkt_assembly_upload(mi355x_kkt_handle h, int seg);
int  mi355x_kkt_factor_assembled(mi355x_kkt_handle h, const double* scale, const double* shift, int* num_neg, int* num_zero);

/* ---- the 8-block primal-dual system on the device -------------------------------------------------------
 * What PDFullSpaceSolver does around the augmented-system solves on host vectors -- eliminate the bound rows into the
 * right-hand side and expand the solution again (SolveOnce, IpPDFullSpaceSolver.cpp:418-424,653-659), form the residual of the
 * UNREDUCED system (ComputeResiduals, :666-793) and its max norms (ComputeResidualRatio, :795-820) for iterative refinement --
 * with the vectors resident in device memory: one upload of the right-hand side and one download of the result per Solve
 * instead of two PCIe round trips and two single-threaded host sparse products per refinement step.  A primal-dual vector is
 * the concatenation x | s | y_c | y_d | z_L | z_U | v_L | v_U (IteratesVector's component order); the handle owns
 * MI355X_KKT_PD_NVEC of them.  W, J_c and J_d are read from the device-resident sources of the value assembly (segments
 * `segs`), the expansion matrices P are their index lists (ExpansionMatrix::ExpandedPosIndices, 0-based), the iterate's
 * multipliers and slacks are uploaded with _pd_put_data.  The inertia-correction loop, the choice of the perturbations and
 * the refinement control stay with the caller (ipopt_adapter/IpMi355xPDSystemSolver.cpp).
 *   dims8   = { n_x, n_s, n_c, n_d, n_xL, n_xU, n_sL, n_sU }          (n_x + n_s + n_c + n_d = the analysed dimension)
 *   data8   = { z_L, z_U, v_L, v_U, slack_x_L, slack_x_U, slack_s_L, slack_s_U }
 *   _pd_solve_once: res <- alpha sol + beta res with sol the solution for right-hand side `rhs` through the CURRENT factorisation
 *   _pd_residual:   resid <- K8 res - rhs with the perturbations deltas4 = { delta_x, delta_s, delta_c, delta_d };
 *                   norms3 = max norms of rhs, res, resid */
#define MI355X_KKT_PD_NVEC 4
int  mi355x_kkt_pd_define(mi355x_kkt_handle h, const int32_t* dims8, const int32_t* idx_xl, const int32_t* idx_xu, const int32_t* idx_sl,
                          const int32_t* idx_su, const int32_t* irn, const int32_t* jcn, const int32_t* segs, int nsegs);
int  mi355x_kkt_pd_put_data(mi355x_kkt_handle h, const double* const* data8);
int  mi355x_kkt_pd_put(mi355x_kkt_handle h, int vec, const double* const* blocks8);
int  mi355x_kkt_pd_get(mi355x_kkt_handle h, int vec, double* const* blocks8);
int  mi355x_kkt_pd_solve_once(mi355x_kkt_handle h, int rhs, int res, double alpha, double beta);
int  mi355x_kkt_pd_residual(mi355x_kkt_handle h, int rhs, int res, int resid, const double* deltas4, double* norms3);

/* Solve A X = B in place for nrhs right-hand sides, rhs[irhs*ld + i], host memory. */
int  mi355x_kkt_solve(mi355x_kkt_handle h, int nrhs, double* rhs_inout, int ld);
/* Same with rhs/solution resident in device memory (nrhs columns, leading dimension ld). */
int  mi355x_kkt_solve_device(mi355x_kkt_handle h, int nrhs, double* d_rhs_inout, int ld);
/* Out-of-place variant: X = A^{-1} B, B untouched.  All device buffers handed to the library must be
 * complete when the call is made (the library works on its own stream; the caller synchronises the
 * stream that produced them). */
int  mi355x_kkt_solve_device2(mi355x_kkt_handle h, int nrhs, const double* d_b, int ldb, double* d_x, int ldx);

int  mi355x_kkt_set_pivtol(mi355x_kkt_handle h, double u);
int  mi355x_kkt_set_pivtolmax(mi355x_kkt_handle h, double umax);
/* IncreaseQuality (IpSparseSymLinearSolverInterface.hpp:220): raises u <- min(pivtolmax, u^0.75) (the rule of
 * IpMa97SolverInterface.cpp:822-854, IpMa27TSolverInterface.cpp:724-740) and returns 1 -- the caller then refactors,
 * mi355x_kkt_refactor -- or returns 0 when u is at its maximum (with opts.smart_quality = 1 also when the last factorisation found
 * that no pivot decision depends on u up to pivtolmax: nothing a refactorisation could improve).  *new_u (may be NULL) receives the new u. */
int  mi355x_kkt_increase_quality(mi355x_kkt_handle h, double* new_u);
int  mi355x_kkt_get_info(mi355x_kkt_handle h, mi355x_kkt_info* info);
/* Symmetric scaling at run time (the option `scaling` only sets the initial mode): 0 none, 1 Ruiz inf-norm equilibration
 * on the device (the algorithm of MC77), 2 the caller's factors (n doubles, caller's numbering; copied), 3 maximum-product
 * matching scaling (the job of MC64: Duff & Koster 2001; host algorithm, recomputed at every factorisation while selected), 4 the same computed once and
 * reused until mi355x_kkt_increase_quality or a new _set_scaling (IpMa97SolverInterface.cpp:725-771: SWITCH_AT_START_REUSE / ON_DEMAND_REUSE), 5 / 6 = the job of
 * 3 / 4 done on the device by a Jacobi auction (feasible duals by construction: |s_i a_ij s_j| <= 1 on every entry; the matched entries of the unsymmetrised
 * scaling are >= e^(-1/64); info.matching_ms / _rounds / _unmatched describe the last computation).  _get_scaling returns the factors the last
 * factorisation used.  Together they give the MA97 call protocol its meaning: control.scaling > 0 => compute and hand back in
 * scale[], control.scaling == 0 with scale != NULL => reuse the caller-held factors, else none (IpMa97SolverInterface.cpp:641-678). */
int  mi355x_kkt_set_scaling(mi355x_kkt_handle h, int mode, const double* user_factors);
int  mi355x_kkt_get_scaling(mi355x_kkt_handle h, double* factors_out);
/* Stand-alone (no handle): Ruiz factors of a triplet matrix computed on the device -- the hook for a host that scales outside
 * the solver, i.e. Ipopt's TSymScalingMethod (IpTSymLinearSolver.cpp:429-441,511-514; cf. IpMc19TSymScalingMethod.cpp:100-204). */
int  mi355x_kkt_ruiz_scaling(int device, int n, int nnz, const int* irn, const int* jcn, const double* a, int index_base,
                             int sweeps, double* factors_out);
/* Stand-alone, HOST (no GPU needed): symmetric maximum-product matching scaling of a triplet matrix -- what HSL's MC64 provides
 * to MA97 / SPRAL (`ma97_scaling mc64`, `spral_scaling matching`): |s_i a_ij s_j| <= 1 with equality on a maximum transversal.
 * *num_unmatched (may be NULL) = structural rank deficiency. */
int  mi355x_kkt_matching_scaling(int n, int nnz, const int* irn, const int* jcn, const double* a, int index_base,
                                 double* factors_out, int* num_unmatched);
/* The columns (caller's index base) whose pivot was numerically zero in the last factorisation, ascending; *count = how
 * many there are (idx may be NULL / shorter).  This is what DetermineDependentRows needs
 * (IpSparseSymLinearSolverInterface.hpp:240-255; MUMPS' PIVNUL_LIST, IpMumpsSolverInterface.cpp:617-709). */
int  mi355x_kkt_zero_pivots(mi355x_kkt_handle h, int* idx, int capacity, int* count);
/* Delayed pivoting across fronts -- what MA27 / MA57 / MA97 / MUMPS / SPRAL do with a fully-summed column that finds no acceptable pivot in
 * its front (Duff & Reid 1983; consequences read at IpMa97SolverInterface.cpp:719-779, IpMa27TSolverInterface.cpp:565-622,
 * IpSpralSolverInterface.cpp:199-204).  factor / refactor / factor_assembled do it themselves (opts.delay_rounds); the two pieces are exposed:
 *   _failed_pivots   the columns (caller's index base, ascending) the last factorisation eliminated although they failed the threshold tests
 *   _delay_columns   move the given columns from their front's supernode to the parent front's and rebuild the symbolic structures (host work;
 *                    a handle with a device sets the numeric side up again; the next factor() uses the new structure).  *moved = columns
 *                    that moved (a root front has no parent); a column that has been moved before climbs 2, 4, 8 ... tree levels. */
int  mi355x_kkt_failed_pivots(mi355x_kkt_handle h, int* idx, int capacity, int* count);
int  mi355x_kkt_delay_columns(mi355x_kkt_handle h, const int* cols, int count, int* moved);
int  mi355x_kkt_set_delay_rounds(mi355x_kkt_handle h, int rounds);      /* opts.delay_rounds at run time (0 = static pivoting) */
const char* mi355x_kkt_last_error(mi355x_kkt_handle h);

/* ---- symbolic introspection (host logic tests, debugging; sizes via get_info) ---- */
/* what: 0 perm[n] (new->old), 1 sn_colptr[num_sn+1], 2 sn_rowptr[num_sn+1], 3 sn_rows[sum_sn_rows],
 *       4 sn_parent[num_sn], 5 sn_level[num_sn], 6 rel[sum_sn_rows] (position of each update row in the
 *       parent's front, -1 for pivot rows), 7 aperm_colptr[n+1], 8 aperm_row[nnz_a] (permuted lower CSC),
 *       9 trip2slot[nnz_in] (triplet -> permuted CSC slot), 10 pair_of[n] (old index of the 2x2 partner or -1),
 *       11 sn_owner[num_sn] (multi-GPU rank owning the supernode, -1 = replicated top),
 *       12 apos[nnz_a] (row + col*m position of each permuted-CSC slot inside its supernode panel),
 *       13 level_ptr[num_levels*4+1], 14 level_sn[num_sn] (launch schedule: buckets (level, front class)),
 *       15 grp_pos[num_sn], 16 grp_rem[num_sn] (chain groups: position in the group, columns of the later links),
 *       17 alias_child[num_sn] (in-place chains: the child whose contribution block hosts this front, or -1),
 *       18 sn_glo[num_sn], 19 sn_gsz[num_sn] (multi-GPU: the range of ranks [glo, glo + gsz) that holds the front: one rank for an owned
 *       front, all ranks for the classic replicated top, the ranks beneath it with opts.subcube), 20 sn_gdepth[num_sn] (bisections of the
 *       machine above that range = the exchange step the front belongs to),
 *       21 dup_ptr[nnz_a + 1], 22 dup_src[nnz_in] (the triplets of every CSC slot in ascending order: the order in which the device sums
 *       duplicates), 23 sn_class[num_sn] (kernel class of the front: 0 order <= 32, 1 <= 64, 2 <= 128, 3 the blocked path, order > 128),
 *       24 rslot_ptr[n + 1], 25 rslot_idx[rslot_ptr[n]], 26 rslot_col[rslot_ptr[n]] (the symmetric row view of the permuted pattern the equilibration
 *       sweeps and the device refinement gather over: for every row its entries of both triangles -- CSC slot and the other index -- by ascending other index).
 *       sn_parent (4) is the parent in the ASSEMBLY tree: a side child of an in-place chain link may hang on a lower link of that chain (finish_analysis 9b).
 *  27   the storage plan of the contribution blocks, 5 ints: {window of tree levels after which a dead block's space is written again (0: every block
 *       resident), then as (low, high) 32-bit halves: doubles of all blocks if every one were resident, doubles of the blocks that are never reused};
 *       info.cb_doubles is what the plan needs.  Blocks that only carry a contribution to their parent are RECYCLED over the level schedule where that
 *       saves a quarter of the pool (MI355X_KKT_RECYCLE=0/1 forces it off / on) -- the workspace a multifrontal code keeps as a stack
 *       (IpMumpsSolverInterface.cpp:151-177, ICNTL(14)), here with static addresses.
 *  28   the offset (doubles, inside the contribution-block arena) of every front's block as (low, high) halves: 2 * num_sn ints. */
int  mi355x_kkt_get_symbolic(mi355x_kkt_handle h, int what, int* out, int64_t capacity);

/* ---- measurement: device time per kernel kind (hip events around every launch of an eager, graph-less factor + one
 * solve -- on the stream the kernel is launched on: the look-ahead parts of the largest trailing updates run, and are
 * measured, on the solver's second stream exactly as in a timed factorisation --, accumulated over `reps` repetitions).
 * ms/launches need MI355X_KKT_KERNEL_COUNT entries.  Used by bench.py for the roofline of the dominant kernel. ---- */
#define MI355X_KKT_KERNEL_GATHER_SCALE  0   /* value gather + Ruiz equilibration                                   */
#define MI355X_KKT_KERNEL_FRONT_WAVE    1   /* k_front_dpp16 + k_front_reg<64,*> : fronts of order <= 32            */
#define MI355X_KKT_KERNEL_FRONT_LDS64   2   /* k_front_reg<64,8>  : order <= 64, one wavefront each                */
#define MI355X_KKT_KERNEL_FRONT_LDS128  3   /* k_front_reg<256,*> : order <= 128                                   */
#define MI355X_KKT_KERNEL_BIG_ASSEMBLE  4
#define MI355X_KKT_KERNEL_BIG_DIAG      5
#define MI355X_KKT_KERNEL_BIG_TRSM      6
#define MI355X_KKT_KERNEL_BIG_SCHUR     7   /* fp64 MFMA frontal update                                            */
#define MI355X_KKT_KERNEL_STATS         8
#define MI355X_KKT_KERNEL_SOLVE_PERM    9
#define MI355X_KKT_KERNEL_FWD_WAVE     10
#define MI355X_KKT_KERNEL_FWD_LDS      11
#define MI355X_KKT_KERNEL_FWD_BIG      12
#define MI355X_KKT_KERNEL_BWD_WAVE     13
#define MI355X_KKT_KERNEL_BWD_LDS      14
#define MI355X_KKT_KERNEL_BWD_BIG      15
#define MI355X_KKT_KERNEL_FWD_BIG_UPD 16   /* chain groups: update of the entries beyond the group (many workgroups) */
#define MI355X_KKT_KERNEL_BWD_BIG_DOT 17   /* chain groups: partial L21^T x of the rows beyond the group             */
#define MI355X_KKT_KERNEL_COUNT        18
int  mi355x_kkt_profile(mi355x_kkt_handle h, int reps, double* ms, int* launches, int capacity);

/* ---- multi-GPU (one process per GPU; subtrees sharded, top of the tree replicated) ----
 * Create every rank's handle with opts.nranks = P, opts.rank = r, analyse the SAME structure on every rank, then give the
 * handle a communicator.  From then on the ordinary entry points (factor / refactor / solve / solve_device*) run the
 * distributed sequence themselves, on the solver's stream, without host synchronisation between the phases:
 *     factor:  own subtrees -> all-reduce(top arena: lower triangles of the join fronts, fp64 sum) -> replicated top -> all-reduce(inertia counters)
 *     solve :  local forward -> all-reduce(top rhs) -> replicated top fwd/bwd -> local backward -> all-reduce(solution)
 * Inputs (values, right-hand sides) are identical on every rank, outputs (inertia, status, solution) too -- which is what
 * Ipopt needs when every rank runs the same (deterministic) algorithm around its share of the linear algebra; cf. the
 * knobs the reference exposes for SPRAL's multi-GPU mode (IpSpralSolverInterface.cpp:55-67).
 *   _comm_unique_id        rank 0: ncclGetUniqueId -> 128 bytes the launcher hands to every rank (file, MPI, torch store ...)
 *   _set_comm_rccl         ncclCommInitRank(nranks, id, rank) on the handle's device: RCCL over xGMI.  librccl.so is
 *                          dlopen()ed here; single-GPU users never load it
 *   _set_comm_callbacks    a host that has its own communication layer supplies the one collective we need: in-place sum over
 *                          the ranks of `count` elements of DEVICE memory (dtype 0 = fp64, 1 = int32), ordered after the work
 *                          already enqueued on `hip_stream` and complete (or stream-ordered) when it returns; 0 = success */
typedef int (*mi355x_kkt_allreduce_fn)(void* ctx, void* dptr, int64_t count, int dtype, void* hip_stream);
int  mi355x_kkt_comm_unique_id(void* out128);
int  mi355x_kkt_set_comm_rccl(mi355x_kkt_handle h, const void* unique_id128);
int  mi355x_kkt_set_comm_callbacks(mi355x_kkt_handle h, mi355x_kkt_allreduce_fn fn, void* ctx);
/*   _comm_shm_id           rank 0: creates a POSIX shared-memory segment for `nranks` ranks of ONE node and writes its name into out128 (handed to
 *                          the other ranks like the RCCL id)
 *   _set_comm_shm          every rank: attaches (rank 0 unlinks the name once all have) and installs a host-staged sum -- device -> slot, sum of
 *                          the slots in rank order (bitwise the same on every rank), -> device -- for both collectives above.  For ranks that
 *                          SHARE a device (RCCL refuses that), bring-up and the one-GPU test box; the production path is RCCL.  A peer that does
 *                          not arrive within MI355X_KKT_SHM_TIMEOUT_S (300) fails the call: never a hang.  The communicator in the adapter has the
 *                          reference's MPI_Init-inside-the-interface as its precedent (IpMumpsSolverInterface.cpp:58-75). */
int  mi355x_kkt_comm_shm_id(void* out128, int nranks);
void mi355x_kkt_comm_shm_discard(const void* id128);      /* rank 0: the id could not be handed to the other ranks after all -- unlink the segment nobody attached to */
int  mi355x_kkt_set_comm_shm(mi355x_kkt_handle h, const void* id128);
/* Range-local exchange.  A replicated front is held by a RANGE of ranks [rank_lo, rank_lo + nranks_in_range) and what it receives comes from
 * ranks of that range only, so its arena square (lower triangle, packed) and its top right-hand side are summed among those ranks alone:
 * with RCCL through one sub-communicator per exchange step (ncclCommSplit, created by _set_comm_rccl; without it -- or with
 * MI355X_KKT_DISABLE=subcomm -- every step is ONE all-reduce over the whole communicator, ranks outside a range contributing zeros), with a
 * callback communicator through this optional second callback (same contract as mi355x_kkt_allreduce_fn, plus the range; ctx is the one
 * given to _set_comm_callbacks; only ranks of the range call it).
 *   _exchange_bytes   bytes of all arena squares / all top right-hand sides of the current structure (what the exchange steps move) */
typedef int (*mi355x_kkt_allreduce_range_fn)(void* ctx, void* dptr, int64_t count, int dtype, void* hip_stream, int rank_lo, int nranks_in_range);
int  mi355x_kkt_set_comm_range_callback(mi355x_kkt_handle h, mi355x_kkt_allreduce_range_fn fn);
int  mi355x_kkt_exchange_bytes(mi355x_kkt_handle h, int64_t* arena_bytes, int64_t* rhs_bytes);
/* What ONE rank of the handle's nranks will ask of its communicator, in order, for one factorisation + one solve -- and the ncclCommSplit calls
 * _set_comm_rccl makes before them.  HOST ONLY (needs _analyse, not a device): a launcher or a test can check on any machine that all ranks
 * issue matching sequences (same communicator, same count, same order) before an 8-GPU job is started; the reference's only counterpart is
 * MUMPS' own MPI layer behind IpMumpsSolverInterface.cpp:191-245.  Records of 6 ints {what, step, colour, range size, count, dtype}:
 *   what 0 = ncclCommSplit(colour, key = rank) (colour -1 = NCCL_SPLIT_NOCOLOR), 1 = sum of arena squares, 2 = inertia / pivot statistics,
 *        3 = sum of top right-hand sides, 4 = sum of the solution pieces;  colour = first rank of the range whose sub-communicator carries the
 *        collective, -2 = the whole communicator;  dtype 0 = fp64, 1 = int32;  range_local = 0: the fall-back of one whole-communicator sum per step.
 *   _comm_info   the communicator in use: kind 0 none / 1 callbacks / 2 RCCL, the size the communicator itself reports (ncclCommCount), whether
 *                range-local collectives are on (ncclCommSplit succeeded everywhere / a range callback is set), exchange steps of the structure */
int  mi355x_kkt_comm_plan(mi355x_kkt_handle h, int rank, int range_local, int* records6, int capacity_records, int* count);
int  mi355x_kkt_comm_info(mi355x_kkt_handle h, int* kind, int* ranks_seen, int* range_local, int* exchange_steps);
/* The phases are also exposed one by one (a caller that wants to overlap or replace the collectives): */
/* The top-of-tree fronts live in one contiguous device buffer ("top arena").  After
 * factor_local() each rank holds its own subtrees' Schur contributions there; the caller
 * sums the arena over ranks (RCCL all-reduce) and calls factor_top().  See DESIGN.md (e). */
int  mi355x_kkt_factor_local(mi355x_kkt_handle h, const double* dvals);
int  mi355x_kkt_top_arena(mi355x_kkt_handle h, double** dptr, int64_t* ndoubles);
int  mi355x_kkt_factor_top(mi355x_kkt_handle h, int* num_neg_local, int* num_zero_local);
int  mi355x_kkt_solve_fwd_local(mi355x_kkt_handle h, double* d_rhs);
int  mi355x_kkt_top_rhs(mi355x_kkt_handle h, double** dptr, int64_t* ndoubles);
int  mi355x_kkt_solve_top_and_bwd(mi355x_kkt_handle h, double* d_rhs);

#ifdef __cplusplus
}
#endif
#endif /* MI355X_KKT_H */

// numeric.h -- device-side numeric engine interface (implemented in numeric.hip).
// Host code (api.cpp) owns a Numeric object per solver handle.
#pragma once
#include "symbolic.h"
#include <cstdint>
#include <string>
#include <vector>

namespace mi355x {

struct NumericOptions {
    int    device = -1;
    int    scaling = 1;
    double pivtol = 1e-8;
    double pivtolmax = 1e-4;   // the largest u IncreaseQuality can reach: the factorisation records whether any pivot decision would differ there
    double small = 1e-20;
    int    refine_steps = 0;
    int    use_graph = 1;
    int    rank = 0, nranks = 1;
    int    verbose = 0;
    void*  prewarmed_vals = nullptr;   // pinned staging buffer of prewarmed_count doubles made by Numeric::prewarm while the analysis ran (setup takes ownership)
    size_t prewarmed_count = 0;
};

// num_small = failed pivots (num_delay of MA97/SSIDS): eliminated although they failed the threshold test, because the static
// structure cannot delay them to the parent front; u_sensitive != 0: some pivot decision would differ at u = pivtolmax
// num_fast = pivot blocks of big fronts accepted on the blocked a-posteriori path (the others took the strict loop)
struct FactorStats { int num_neg = 0, num_zero = 0, num_two = 0, num_small = 0, u_sensitive = 1, num_fast = 0; };

class NumericImpl;

class Numeric {
public:
    Numeric();
    ~Numeric();
    // false + error() on failure (e.g. no HIP device: there is NO CPU fallback)
    bool   setup(const Symbolic& S, const NumericOptions& opt);
    double* values_buffer();                         // pinned host buffer, nnz_in doubles
    bool   factor(const double* dvals_or_null, bool reuse_device_values, FactorStats& st);
    bool   solve_host(int nrhs, double* rhs, int ld);
    bool   solve_device(int nrhs, double* drhs, int ld);
    bool   solve_device2(int nrhs, const double* db, int ldb, double* dx, int ldx);   // out of place
    void   set_pivtol(double u);
    void   set_pivtolmax(double u);
    double last_factor_ms() const;
    void   matching_stats(double* ms, int* rounds, int* unmatched) const;      // the last matching-scaling computation (scaling modes 3-6)
    double last_solve_ms() const;
    const std::string& error() const;
    static constexpr int kNumKernelKinds = 18;
    bool   profile(int reps, double* ms, int* launches);
    bool   debug_pivots(double* dinv, double* doff, int* ptype, int* lperm);      // development aid: pivot data of the last factorisation, n entries each
    bool   debug_clocks(unsigned long long* out128);        // development aid (MI355X_KKT_TRACE=clocks)   // per-kernel-kind device time (hip events), eager launches
    // multi-GPU pieces
    bool   factor_local(const double* dvals_or_null);
    bool   top_arena(double** dptr, int64_t* ndoubles);
    bool   factor_top(FactorStats& st);
    bool   solve_fwd_local(double* drhs);
    bool   top_rhs(double** dptr, int64_t* ndoubles);
    bool   solve_top_and_bwd(double* drhs);
    bool   set_scaling(int mode, const double* user_factors_orig_numbering);   // 0 none, 1 Ruiz (device), 2 the caller's factors, 3 matching (host, every factorisation), 4 matching (host, computed once, reused), 5 / 6 the same on the device
    bool   get_scaling(double* out_orig_numbering);                             // factors of the last factorisation
    void   invalidate_matching();                                                // scaling mode 4: compute the matching scaling afresh at the next factorisation
    // first touch of the device (context, code objects) and the pinned staging buffer: independent of the analysis, so the C API runs it on a
    // thread next to it (0.1-0.3 s of an Ipopt run's LinearSystemSymbolicFactorization otherwise).  Returns the buffer or nullptr.
    static int   resolve_device(int device);              // on the caller's thread: the ordinal `device` (or the current device for -1) means, -1 without a device
    static void* prewarm(int device, size_t count);
    static void prewarm_discard(void* p);
    static bool ruiz_triplet(int device, int n, int nnz, const int* irn, const int* jcn, const double* a, int base, int sweeps, double* out, std::string& err);
    bool   zero_pivots(std::vector<int>& idx0);           // columns (original numbering, 0-based) with a zero pivot in the last factorisation
    // delayed pivoting across fronts: the columns (CURRENT permuted numbering) the last factorisation could not pivot, and the re-setup of
    // everything that depends on the elimination structure once symbolic.cpp has moved them to their parent fronts (values, assembly and
    // primal-dual state, scaling factors, communicator and the pinned buffers handed to the caller survive)
    bool   failed_pivots(std::vector<int>& cols_perm);
    bool   restructure(const Symbolic& S);
    // device-side value assembly: triplet values = concatenated segments, each  scale * src + shift  from a device-resident source
    bool   assembly_define(int nseg, const int64_t* off, const int64_t* len);
    double* assembly_buffer(int seg);                      // pinned staging of the segment's source values
    bool   assembly_upload(int seg);                       // async H2D on the solver's stream
    bool   factor_assembled(const double* scale, const double* shift, FactorStats& st);
    // the 8-block primal-dual system on the device: vectors are {x, s, y_c, y_d, z_L, z_U, v_L, v_U} concatenated
    bool   pd_define(const int* dims8, const int* ixl, const int* ixu, const int* isl, const int* isu, const int* irn, const int* jcn, const int* segs, int nsegs);
    bool   pd_put_data(const double* const* arr8);
    bool   pd_put(int vec, const double* const* blocks8);
    bool   pd_get(int vec, double* const* blocks8);
    bool   pd_solve_once(int rhs, int res, double alpha, double beta);
    bool   pd_residual(int rhs, int res, int resid, const double* deltas4, double* norms3);
    // communicator of a multi-GPU handle: with one set, factor()/solve_*() run the whole distributed sequence themselves
    bool   set_comm_rccl(const void* unique_id128);                                   // RCCL (dlopen'ed), ncclCommInitRank(nranks, id, rank)
    bool   set_comm_callback(int (*allreduce)(void* ctx, void* dptr, int64_t count, int dtype, void* hip_stream), void* ctx);
    // optional, for a callback communicator: in-place sum over the ranks [rank_lo, rank_lo + nranks_in_range) only (this rank is one of them);
    // with it a range of ranks sums its part of an exchange step among itself instead of the whole machine summing everything
    bool   set_comm_range_callback(int (*allreduce_range)(void* ctx, void* dptr, int64_t count, int dtype, void* hip_stream, int rank_lo, int nranks_in_range));
    long long exchange_bytes(int what) const;        // bytes of the arena squares (0) / top right-hand sides (1) of all exchange steps
    // host only (no device): the collectives rank `rank` of `nranks` issues for one factorisation + one solve of the structure S, and the ncclCommSplit calls
    // before them -- records of 6 ints, see NumericImpl::comm_plan.  What a CPU test walks to show that every rank issues matching sequences.
    static void comm_plan(const Symbolic& S, int nranks, int rank, bool range_local, std::vector<int>& out6);
    // the communicator in use: kind 0 none / 1 callbacks / 2 RCCL; ranks the communicator itself reports (ncclCommCount; nranks for callbacks); range-local collectives on
    void comm_info(int* kind, int* ranks_seen, int* range_local, int* exchange_steps) const;
    static bool rccl_unique_id(void* out128, std::string& err);                       // ncclGetUniqueId (rank 0 creates, the launcher distributes)
private:
    NumericImpl* p_;
};

} // namespace mi355x

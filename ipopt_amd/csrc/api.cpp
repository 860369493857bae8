// api.cpp -- the C ABI of include/mi355x_kkt.h on top of Symbolic (host) + Numeric (HIP).
// No exception may cross this boundary ("never let one cross our C ABI").
#include "../../include/mi355x_kkt.h"
#include "symbolic.h"
#include "numeric.h"
#include "matching_scaling.h"
#include "comm_shm.h"
#include "env_knobs.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <chrono>
#include <vector>

using namespace mi355x;

struct mi355x_kkt_handle_s {
    mi355x_kkt_options opts;
    Symbolic sym;
    Numeric* num = nullptr;
    bool analysed = false, numeric_ready = false, factored = false;
    bool stats_stale = false;      // the scaling mode / factors changed since the factorisation `last` describes: its u_sensitive verdict no longer applies
    FactorStats last;
    SymbolicOptions so;            // what analyse() ran with: the structure is edited again with the same options when pivots are delayed
    int64_t base_nnz_l = 0;        // nnz(L) of the analysis (the delayed-pivot edits may not grow the factor without bound)
    int num_delayed = 0;           // columns moved to a parent front since analyse() (a column that moved twice counts twice), MA97's num_delay
    int num_restructures = 0;      // structure edits since analyse()
    std::vector<unsigned char> delay_count;   // per column (caller's numbering, 0-based): how often it has been moved up
    // latches of the delayed-pivot loop, cleared by analyse():
    bool delays_exhausted = false; // an edit ran into the growth cap (or nothing could move any more): static pivoting from here on, no more edits are built
    int  zero_futile_events = 0;   // consecutive rounds driven by ZERO pivots alone that did not lower their number (latched at 2: see zero_delay_futile)
    bool zero_delay_futile = false;// a round driven by ZERO pivots alone did not lower their number: the matrix is singular (a dependent row is a zero pivot
                                   // wherever it is eliminated) -- later factorisations whose only complaint is zero pivots answer SINGULAR at once
    std::thread reaper;            // destroys the structure a delayed-pivot edit replaced (80-90 ms of unmapping at n = 10^6) off the caller's path
    ShmComm* shm = nullptr;        // the shared-memory communicator of set_comm_shm (owned; the Numeric object only holds the callbacks' context)
    std::string err;
    std::string setup_err;         // why the device set-up of the last analyse() failed (no device; the pool does not fit ...): every later "not ready" answer repeats it
    std::vector<double> host_vals_nodev;   // plain host staging buffer handed out when no device exists (values only, never computed on)
};

extern "C" {

void mi355x_kkt_default_options(mi355x_kkt_options* o)
{
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->device = -1; o->index_base = 1; o->ordering = 0; o->matching = 1; o->scaling = 1;
    o->nd_leaf = 32; o->nemin = 8; o->max_sn_cols = 64;
    o->pivtol = 1e-8; o->pivtolmax = 1e-4; o->small = 1e-20;
    o->refine_steps = 0; o->use_graph = 1; o->nranks = 1; o->rank = 0; o->verbose = 0; o->leaf_cols = 0; o->tree_merge = 0; o->wide_panels = 0; o->chain_group = 4; o->solve_group = 0; o->subcube = 0;
    o->delay_rounds = 8; o->smart_quality = 0;
}

int mi355x_kkt_create(mi355x_kkt_handle* h, const mi355x_kkt_options* opts)
{
    if (!h) return MI355X_KKT_FATAL;
    try {
        auto* p = new mi355x_kkt_handle_s();
        if (opts) p->opts = *opts; else mi355x_kkt_default_options(&p->opts);
        *h = p; return MI355X_KKT_SUCCESS;
    } catch (...) { *h = nullptr; return MI355X_KKT_FATAL; }
}

void mi355x_kkt_destroy(mi355x_kkt_handle h)
{
    if (!h) return;
    try { if (h->reaper.joinable()) h->reaper.join(); delete h->num; shm_comm_destroy(h->shm); delete h; } catch (...) {}
}

const char* mi355x_kkt_last_error(mi355x_kkt_handle h) { return h ? h->err.c_str() : "null handle"; }

int mi355x_kkt_analyse(mi355x_kkt_handle h, int n, int nnz, const int* row, const int* col, int format, const double* vals)
{
    if (!h) return MI355X_KKT_FATAL;
    try {
        h->analysed = false; h->numeric_ready = false; h->factored = false;
        if (h->reaper.joinable()) h->reaper.join();
        delete h->num; h->num = nullptr;
        h->num_delayed = 0; h->num_restructures = 0; h->delay_count.clear(); h->delays_exhausted = false; h->zero_delay_futile = false; h->zero_futile_events = 0;
        SymbolicOptions& so = h->so; so = SymbolicOptions();
        so.index_base = h->opts.index_base; so.ordering = h->opts.ordering; so.matching = h->opts.matching;
        so.nd_leaf = h->opts.nd_leaf > 0 ? h->opts.nd_leaf : 32; so.nemin = h->opts.nemin > 0 ? h->opts.nemin : 8;
        so.max_sn_cols = h->opts.max_sn_cols > 1 ? (h->opts.max_sn_cols > 64 ? 64 : h->opts.max_sn_cols) : 64;   // 64: LDS budget of k_big_trsm (104 KiB at k = 65)
        so.nranks = h->opts.nranks > 0 ? h->opts.nranks : 1; so.verbose = h->opts.verbose; so.leaf_cols = h->opts.leaf_cols; so.tree_merge = h->opts.tree_merge;
        so.wide_panels = h->opts.wide_panels; so.chain_purify = knob_disabled("purify") ? 0 : 1; so.chain_group = h->opts.chain_group > 0 ? h->opts.chain_group : 4; so.solve_group = h->opts.solve_group; so.subcube = h->opts.subcube;
        // The device's first touch (runtime, context, code objects) and the pinned staging buffer do not depend on the analysis: the two run side by side.
        // HIP's "current device" belongs to the CALLING thread (a helper thread asking for it gets device 0 whatever the caller selected), and finding it
        // out initialises the runtime -- a good part of what is to be overlapped -- so the CALLER does the warm-up and the helper thread the analysis,
        // which is plain host code.  The thread is joined on every path out of this scope, exceptions included (a joinable std::thread that is destroyed
        // terminates the process); what it threw comes back as its error.
        auto wall_ = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t_in = wall_();
        void* pre = nullptr;
        bool aok = false; std::string aerr;
        {
            struct Joiner { std::thread t; ~Joiner() { if (t.joinable()) t.join(); } } an;
            an.t = std::thread([&] {
                try { aok = analyse(h->sym, so, n, nnz, row, col, format, vals); if (!aok) aerr = h->sym.error;
                      if (h->opts.verbose >= 2) fprintf(stderr, "[mi355x_kkt]   analyse() returned %.3f s after the call\n", wall_() - t_in); }
                catch (const std::bad_alloc&) { aok = false; aerr = "analyse: out of host memory"; }
                catch (...) { aok = false; aerr = "analyse: unexpected exception"; }
            });
            try { pre = Numeric::prewarm(h->opts.device, (size_t)(nnz > 0 ? nnz : 1)); } catch (...) { pre = nullptr; }
            if (h->opts.verbose >= 2) fprintf(stderr, "[mi355x_kkt]   device warm-up done %.3f s after the call\n", wall_() - t_in);
        }
        struct PreGuard { void*& p; bool taken; ~PreGuard() { if (!taken && p) { Numeric::prewarm_discard(p); p = nullptr; } } } pg{pre, false};      // (released on every path on which setup() does not take it)
        if (h->opts.verbose >= 2) fprintf(stderr, "[mi355x_kkt]   analysis thread and device warm-up joined %.3f s after the call\n", wall_() - t_in);
        if (!aok) { h->err = aerr; return MI355X_KKT_FATAL; }
        h->analysed = true; h->base_nnz_l = h->sym.nnz_l;
        // device setup is attempted right away so that values_buffer() can hand out pinned memory;
        // without a GPU the symbolic result stays queryable and factor()/solve() fail loudly.
        h->num = new Numeric();
        NumericOptions no;
        no.device = h->opts.device; no.scaling = h->opts.scaling; no.pivtol = h->opts.pivtol; no.small = h->opts.small;
        no.pivtolmax = h->opts.pivtolmax > h->opts.pivtol ? h->opts.pivtolmax : h->opts.pivtol;
        no.refine_steps = h->opts.refine_steps; no.use_graph = h->opts.use_graph; no.rank = h->opts.rank; no.nranks = so.nranks;
        no.verbose = h->opts.verbose;
        no.prewarmed_vals = pre; no.prewarmed_count = (size_t)(nnz > 0 ? nnz : 1);
        pg.taken = true;                       // (setup owns the buffer from here on, whether it succeeds or not)
        h->numeric_ready = h->num->setup(h->sym, no);
        h->setup_err = h->numeric_ready ? std::string() : h->num->error();
        if (!h->numeric_ready) h->err = h->setup_err;
        if (h->opts.verbose >= 2) fprintf(stderr, "[mi355x_kkt]   device set-up done %.3f s after the call\n", wall_() - t_in);
        return MI355X_KKT_SUCCESS;
    } catch (const std::bad_alloc&) { h->err = "analyse: out of host memory"; return MI355X_KKT_FATAL; }
    catch (...) { h->err = "analyse: unexpected exception"; return MI355X_KKT_FATAL; }
}

double* mi355x_kkt_values_buffer(mi355x_kkt_handle h)
{
    if (!h || !h->analysed) return nullptr;
    if (h->numeric_ready) return h->num->values_buffer();
    try { h->host_vals_nodev.resize(h->sym.nnz_in > 0 ? h->sym.nnz_in : 1); return h->host_vals_nodev.data(); } catch (...) { return nullptr; }
}

// Delayed pivoting across fronts (the behaviour of MA27 / MA57 / MA97 / MUMPS / SPRAL whose consequences the reference adapters read:
// IpMa97SolverInterface.cpp:719-779 info.num_delay, IpMa27TSolverInterface.cpp:565-622 "grow the workspace and factor again"): when the
// factorisation had to eliminate columns that failed the threshold test -- every candidate of their front had failed, or a multiplier
// below a pivot block came out above 1/u -- those columns are moved to the parent front's supernode (symbolic.cpp restructure_delays),
// everything that depends on the structure is set up again and the SAME values are refactored; up to opts.delay_rounds times per call.
// The edited structure stays for the following factorisations.  0 rounds = static pivoting (failed pivots forced and counted in num_small).
static bool apply_delays(mi355x_kkt_handle h, const std::vector<int>& marks_perm, int* moved)
{
    Symbolic ns;
    *moved = 0;
    if (marks_perm.empty()) return false;
    // a column that has been delayed before and fails again climbs 2, 4, 8 ... levels: along a separator chain the rows that hold its large
    // entries become fully summed many links further up, and every visit of a link on the way costs a refactorisation
    if ((int)h->delay_count.size() != h->sym.n) h->delay_count.assign(h->sym.n, 0);
    std::vector<int> hops(marks_perm.size());
    for (size_t q = 0; q < marks_perm.size(); ++q) hops[q] = 1 << std::min<int>(h->delay_count[h->sym.perm[marks_perm[q]]], 6);
    std::vector<char> acted;
    const auto t_a = std::chrono::steady_clock::now();
    if (!restructure_delays(h->sym, h->so, marks_perm, hops, ns, moved, &acted)) return false;
    const auto t_b = std::chrono::steady_clock::now();
    if (ns.nnz_l > 4 * h->base_nnz_l + 4000000) {      // the factor may grow, not explode: static pivoting from here on -- and no more edits are BUILT
        *moved = 0; h->delays_exhausted = true;         // (a complete host re-analysis each, 0.1-1 s at n = 10^6, only to be thrown away: ADVICE r04)
        if (h->opts.verbose) fprintf(stderr, "[mi355x_kkt] factor: the delayed-pivot edit would grow nnz(L) to %lld (> 4 x %lld + 4e6): static pivoting from here on\n", (long long)ns.nnz_l, (long long)h->base_nnz_l);
        return false;
    }
    for (size_t q = 0; q < marks_perm.size(); ++q) if (acted[q]) { unsigned char& c = h->delay_count[h->sym.perm[marks_perm[q]]]; if (c < 255) ++c; }
    {   // the old structure goes to a helper thread (joined before the next edit, before a new analysis and at destroy)
        if (h->reaper.joinable()) h->reaper.join();
        Symbolic* old = new Symbolic(std::move(h->sym));
        h->sym = std::move(ns);
        h->reaper = std::thread([old] { delete old; });
    }
    if (h->opts.verbose >= 2) fprintf(stderr, "[mi355x_kkt]   (delay) restructure_delays %.3f s, old structure dropped %.3f s\n", std::chrono::duration<double>(t_b - t_a).count(),
                                       std::chrono::duration<double>(std::chrono::steady_clock::now() - t_b).count());
    h->num_delayed += *moved; h->num_restructures++;
    return true;
}
static bool delay_and_refactor(mi355x_kkt_handle h, FactorStats& st)
{
    // (a forced candidate with nothing usable in its column is counted as a ZERO pivot, not in num_small: both can carry marks -- the hostile
    // band system's 234-270 static zero pivots all disappear once their columns have moved up.)  But a SINGULAR matrix -- Ipopt's rank-deficient
    // Jacobian before the delta_c perturbation -- has zero pivots wherever its dependent rows are eliminated: a round that was driven by zero
    // pivots alone and did not lower their number ends the loop and is remembered for the handle (zero_delay_futile), so that every later SINGULAR
    // answer of the run costs one factorisation, not up to delay_rounds re-analyses with their fill.
    if (h->delays_exhausted) return true;
    for (int round = 0; round < h->opts.delay_rounds && (st.num_small > 0 || (st.num_zero > 0 && !h->zero_delay_futile)); ++round) {
        const bool zero_only = st.num_small == 0;
        const int zero_before = st.num_zero;
        std::vector<int> marks; int moved = 0;
        if (!h->num->failed_pivots(marks)) { h->err = h->num->error(); return false; }
        if (!apply_delays(h, marks, &moved)) break;                      // nothing that can move (root fronts) or the growth cap: keep the static result
        if (!h->num->restructure(h->sym)) { h->err = h->num->error(); h->numeric_ready = false; return false; }
        if (h->opts.verbose) fprintf(stderr, "[mi355x_kkt] factor: %d failed + %d zero pivots, %d columns delayed to their parent fronts (round %d), refactoring\n", st.num_small, st.num_zero, moved, round + 1);
        if (!h->num->factor(nullptr, true, st)) { h->err = h->num->error(); return false; }
        if (zero_only && st.num_small == 0 && st.num_zero < zero_before) h->zero_futile_events = 0;
        if (zero_only && st.num_small == 0 && st.num_zero >= zero_before) {
            // one futile round ends THIS call's loop; the latch for the handle needs two in a row (a delayed column's hop doubles each time it
            // fails again, so the round after a futile one may still succeed: ADVICE r05) and is cleared by set_pivtol / set_delay_rounds / delay_columns
            if (++h->zero_futile_events >= 2) h->zero_delay_futile = true;
            if (h->opts.verbose) fprintf(stderr, "[mi355x_kkt] factor: %d zero pivots stay after the delay (singular matrix): no further delays for zero pivots on this structure\n", st.num_zero);
            break;
        }
    }
    return true;
}

static int do_factor(mi355x_kkt_handle h, const double* dvals, bool reuse, int* num_neg, int* num_zero)
{
    if (!h) return MI355X_KKT_FATAL;
    if (!h->analysed) { h->err = "factor: analyse() has not been called"; return MI355X_KKT_FATAL; }
    if (!h->numeric_ready) { if (h->err.empty()) h->err = "factor: no usable HIP device (no CPU fallback)"; return MI355X_KKT_FATAL; }
    try {
        if (h->sym.n == 0) { h->last = FactorStats(); h->factored = true; if (num_neg) *num_neg = 0; if (num_zero) *num_zero = 0; return MI355X_KKT_SUCCESS; }
        FactorStats st;
        if (!h->num->factor(dvals, reuse, st)) { h->err = h->num->error(); return MI355X_KKT_FATAL; }
        if (!delay_and_refactor(h, st)) return MI355X_KKT_FATAL;
        h->last = st; h->factored = true; h->stats_stale = false;
        if (num_neg) *num_neg = st.num_neg;
        if (num_zero) *num_zero = st.num_zero;
        return st.num_zero > 0 ? MI355X_KKT_SINGULAR : MI355X_KKT_SUCCESS;
    } catch (...) { h->err = "factor: unexpected exception"; return MI355X_KKT_FATAL; }
}

int mi355x_kkt_factor(mi355x_kkt_handle h, const double* dvals, int* num_neg, int* num_zero) { return do_factor(h, dvals, false, num_neg, num_zero); }
int mi355x_kkt_refactor(mi355x_kkt_handle h, int* num_neg, int* num_zero) { return do_factor(h, nullptr, true, num_neg, num_zero); }

/* symmetric scaling at run time: what the MA97 call protocol needs (control.scaling / scale[], IpMa97SolverInterface.cpp:641-678) */
int mi355x_kkt_set_scaling(mi355x_kkt_handle h, int mode, const double* user_factors)
{
    if (!h) return MI355X_KKT_FATAL;
    if (!h->numeric_ready) { h->opts.scaling = mode == 2 ? 1 : mode; h->err = std::string("set_scaling: no device set-up") + (h->setup_err.empty() ? " (analyse() first, and a usable HIP device; no CPU fallback)" : ": " + h->setup_err); return MI355X_KKT_FATAL; }
    try { if (!h->num->set_scaling(mode, user_factors)) { h->err = h->num->error(); return MI355X_KKT_FATAL; } h->opts.scaling = mode; h->stats_stale = true; return MI355X_KKT_SUCCESS; } catch (...) { return MI355X_KKT_FATAL; }
}
int mi355x_kkt_get_scaling(mi355x_kkt_handle h, double* out)
{
    if (!h || !out) return MI355X_KKT_FATAL;
    if (!h->numeric_ready || !h->factored) { h->err = "get_scaling: no factorisation available"; return MI355X_KKT_FATAL; }
    try { if (!h->num->get_scaling(out)) { h->err = h->num->error(); return MI355X_KKT_FATAL; } return MI355X_KKT_SUCCESS; } catch (...) { return MI355X_KKT_FATAL; }
}
/* stand-alone: symmetric Ruiz inf-norm equilibration factors of a triplet matrix, computed on the device (no handle needed) */
int mi355x_kkt_ruiz_scaling(int device, int n, int nnz, const int* irn, const int* jcn, const double* a, int index_base, int sweeps, double* factors)
{
    if (n < 0 || nnz < 0 || !factors || (nnz > 0 && (!irn || !jcn || !a))) return MI355X_KKT_FATAL;
    try { std::string err; return Numeric::ruiz_triplet(device, n, nnz, irn, jcn, a, index_base, sweeps > 0 ? sweeps : 4, factors, err) ? MI355X_KKT_SUCCESS : MI355X_KKT_FATAL; }
    catch (...) { return MI355X_KKT_FATAL; }
}

/* stand-alone, HOST: maximum-product matching scaling (the job of MC64) of a symmetric triplet matrix; duplicates are summed,
 * either triangle accepted.  Works without a GPU (it is an analysis-type algorithm like the ordering). */
int mi355x_kkt_matching_scaling(int n, int nnz, const int* irn, const int* jcn, const double* a, int index_base, double* factors, int* num_unmatched)
{
    if (n < 0 || nnz < 0 || !factors || (nnz > 0 && (!irn || !jcn || !a))) return MI355X_KKT_FATAL;
    try {
        // the full symmetric pattern by columns (both triangles), duplicates summed: bucket the entries by column, sort every (short) column by
        // row, merge equal rows -- in triplet order inside a column, so the sums do not depend on anything but the input
        for (int t = 0; t < nnz; ++t) { const int i = irn[t] - index_base, j = jcn[t] - index_base; if (i < 0 || j < 0 || i >= n || j >= n) return MI355X_KKT_FATAL; }
        std::vector<int> cnt((size_t)n + 1, 0);
        for (int t = 0; t < nnz; ++t) { const int i = irn[t] - index_base, j = jcn[t] - index_base; cnt[j + 1]++; if (i != j) cnt[i + 1]++; }
        for (int j = 0; j < n; ++j) cnt[j + 1] += cnt[j];
        std::vector<int> rows(cnt[n]), fill(cnt.begin(), cnt.end() - 1); std::vector<double> vals(cnt[n]);
        for (int t = 0; t < nnz; ++t) {
            const int i = irn[t] - index_base, j = jcn[t] - index_base;
            rows[fill[j]] = i; vals[fill[j]++] = a[t];
            if (i != j) { rows[fill[i]] = j; vals[fill[i]++] = a[t]; }
        }
        std::vector<int> ptr((size_t)n + 1, 0), idx; std::vector<double> av;
        idx.reserve(cnt[n]); av.reserve(cnt[n]);
        std::vector<std::pair<int, double>> col;
        for (int j = 0; j < n; ++j) {
            col.clear();
            for (int p = cnt[j]; p < cnt[j + 1]; ++p) col.emplace_back(rows[p], vals[p]);
            std::stable_sort(col.begin(), col.end(), [](const std::pair<int, double>& x, const std::pair<int, double>& y) { return x.first < y.first; });
            for (size_t q = 0; q < col.size(); ++q) {
                if (q > 0 && col[q].first == col[q - 1].first) av.back() += col[q].second;
                else { idx.push_back(col[q].first); av.push_back(col[q].second); }
            }
            ptr[j + 1] = (int)idx.size();
        }
        for (double& x : av) x = std::fabs(x);
        return matching_scaling(n, ptr.data(), idx.data(), av.data(), factors, num_unmatched) ? MI355X_KKT_SUCCESS : MI355X_KKT_FATAL;
    } catch (...) { return MI355X_KKT_FATAL; }
}

/* DetermineDependentRows support (IpSparseSymLinearSolverInterface.hpp:240-255; MUMPS' null-pivot list,
 * IpMumpsSolverInterface.cpp:617-709): the columns whose pivot was numerically zero in the last factorisation */
int mi355x_kkt_zero_pivots(mi355x_kkt_handle h, int* idx, int capacity, int* count)
{
    if (!h || !count) return MI355X_KKT_FATAL;
    if (!h->factored || !h->numeric_ready) { h->err = "zero_pivots: no factorisation available"; return MI355X_KKT_FATAL; }
    try {
        std::vector<int> z;
        if (h->sym.n > 0 && !h->num->zero_pivots(z)) { h->err = h->num->error(); return MI355X_KKT_FATAL; }
        *count = (int)z.size();
        if (idx) for (int i = 0; i < (int)z.size() && i < capacity; ++i) idx[i] = z[i] + h->opts.index_base;
        return MI355X_KKT_SUCCESS;
    } catch (...) { h->err = "zero_pivots: unexpected exception"; return MI355X_KKT_FATAL; }
}

/* the columns the last factorisation eliminated although they failed the threshold test (after the delayed-pivot rounds: what is left) */
int mi355x_kkt_failed_pivots(mi355x_kkt_handle h, int* idx, int capacity, int* count)
{
    if (!h || !count) return MI355X_KKT_FATAL;
    if (!h->factored || !h->numeric_ready) { h->err = "failed_pivots: no factorisation available"; return MI355X_KKT_FATAL; }
    try {
        std::vector<int> z;
        if (h->sym.n > 0 && !h->num->failed_pivots(z)) { h->err = h->num->error(); return MI355X_KKT_FATAL; }
        for (int& c : z) c = h->sym.perm[c];
        std::sort(z.begin(), z.end());
        *count = (int)z.size();
        if (idx) for (int i = 0; i < (int)z.size() && i < capacity; ++i) idx[i] = z[i] + h->opts.index_base;
        return MI355X_KKT_SUCCESS;
    } catch (...) { h->err = "failed_pivots: unexpected exception"; return MI355X_KKT_FATAL; }
}
/* the structural edit itself, driven by the caller: host work only (the device side follows when there is one) */
int mi355x_kkt_delay_columns(mi355x_kkt_handle h, const int* cols, int count, int* moved)
{
    if (!h || (count > 0 && !cols)) return MI355X_KKT_FATAL;
    if (!h->analysed) { h->err = "delay_columns: analyse() has not been called"; return MI355X_KKT_FATAL; }
    try {
        std::vector<int> marks; marks.reserve(count > 0 ? count : 0);
        for (int i = 0; i < count; ++i) { const int c = cols[i] - h->opts.index_base; if (c < 0 || c >= h->sym.n) { h->err = "delay_columns: index out of range"; return MI355X_KKT_FATAL; } marks.push_back(h->sym.iperm[c]); }
        int mv = 0;
        const bool ok = apply_delays(h, marks, &mv);
        if (moved) *moved = mv;
        if (ok) { h->factored = false; if (h->numeric_ready && !h->num->restructure(h->sym)) { h->err = h->num->error(); h->numeric_ready = false; return MI355X_KKT_FATAL; } }
        return MI355X_KKT_SUCCESS;
    } catch (const std::bad_alloc&) { h->err = "delay_columns: out of host memory"; return MI355X_KKT_FATAL; }
    catch (...) { h->err = "delay_columns: unexpected exception"; return MI355X_KKT_FATAL; }
}

// ---- device-side value assembly ----
int mi355x_kkt_assembly_define(mi355x_kkt_handle h, int nseg, const int64_t* offset, const int64_t* length)
{
    if (!h || !offset || !length) return MI355X_KKT_FATAL;
    if (!h->numeric_ready) { h->err = std::string("assembly_define: no device set-up") + (h->setup_err.empty() ? " (analyse() first, and a usable HIP device; no CPU fallback)" : ": " + h->setup_err); return MI355X_KKT_FATAL; }
    try { if (!h->num->assembly_define(nseg, offset, length)) { h->err = h->num->error(); return MI355X_KKT_FATAL; } return MI355X_KKT_SUCCESS; } catch (...) { return MI355X_KKT_FATAL; }
}
double* mi355x_kkt_assembly_buffer(mi355x_kkt_handle h, int seg)
{
    if (!h || !h->numeric_ready) return nullptr;
    return h->num->assembly_buffer(seg);
}
int mi355x_kkt_assembly_upload(mi355x_kkt_handle h, int seg)
{
    if (!h || !h->numeric_ready) return MI355X_KKT_FATAL;
    try { if (!h->num->assembly_upload(seg)) { h->err = h->num->error(); return MI355X_KKT_FATAL; } return MI355X_KKT_SUCCESS; } catch (...) { return MI355X_KKT_FATAL; }
}
int mi355x_kkt_factor_assembled(mi355x_kkt_handle h, const double* scale, const double* shift, int* num_neg, int* num_zero)
{
    if (!h || !scale || !shift) return MI355X_KKT_FATAL;
    if (!h->analysed || !h->numeric_ready) { h->err = std::string("factor_assembled: no device set-up") + (h->setup_err.empty() ? " (analyse() first, and a usable HIP device; no CPU fallback)" : ": " + h->setup_err); return MI355X_KKT_FATAL; }
    try {
        FactorStats st;
        if (!h->num->factor_assembled(scale, shift, st)) { h->err = h->num->error(); return MI355X_KKT_FATAL; }
        if (!delay_and_refactor(h, st)) return MI355X_KKT_FATAL;
        h->last = st; h->factored = true; h->stats_stale = false;
        if (num_neg) *num_neg = st.num_neg;
        if (num_zero) *num_zero = st.num_zero;
        return st.num_zero > 0 ? MI355X_KKT_SINGULAR : MI355X_KKT_SUCCESS;
    } catch (...) { h->err = "factor_assembled: unexpected exception"; return MI355X_KKT_FATAL; }
}

#define PD_CALL(name, cond, call) \
    if (!h || !(cond)) return MI355X_KKT_FATAL; \
    if (!h->analysed || !h->numeric_ready) { h->err = std::string(name ": no device set-up") + (h->setup_err.empty() ? " (analyse() first, and a usable HIP device; no CPU fallback)" : ": " + h->setup_err); return MI355X_KKT_FATAL; } \
    try { if (!h->num->call) { h->err = h->num->error(); return MI355X_KKT_FATAL; } return MI355X_KKT_SUCCESS; } \
    catch (...) { h->err = name ": unexpected exception"; return MI355X_KKT_FATAL; }
int mi355x_kkt_pd_define(mi355x_kkt_handle h, const int32_t* dims8, const int32_t* ixl, const int32_t* ixu, const int32_t* isl, const int32_t* isu,
                         const int32_t* irn, const int32_t* jcn, const int32_t* segs, int nsegs)
{ PD_CALL("pd_define", dims8 && irn && jcn && segs && nsegs > 0, pd_define(dims8, ixl, ixu, isl, isu, irn, jcn, segs, nsegs)) }
int mi355x_kkt_pd_put_data(mi355x_kkt_handle h, const double* const* data8) { PD_CALL("pd_put_data", data8, pd_put_data(data8)) }
int mi355x_kkt_pd_put(mi355x_kkt_handle h, int vec, const double* const* blocks8) { PD_CALL("pd_put", blocks8, pd_put(vec, blocks8)) }
int mi355x_kkt_pd_get(mi355x_kkt_handle h, int vec, double* const* blocks8) { PD_CALL("pd_get", blocks8, pd_get(vec, blocks8)) }
int mi355x_kkt_pd_solve_once(mi355x_kkt_handle h, int rhs, int res, double alpha, double beta)
{
    if (h && !h->factored) { h->err = "pd_solve_once: no factorisation available"; return MI355X_KKT_FATAL; }
    PD_CALL("pd_solve_once", true, pd_solve_once(rhs, res, alpha, beta))
}
int mi355x_kkt_pd_residual(mi355x_kkt_handle h, int rhs, int res, int resid, const double* deltas4, double* norms3)
{ PD_CALL("pd_residual", deltas4 && norms3, pd_residual(rhs, res, resid, deltas4, norms3)) }
#undef PD_CALL

int mi355x_kkt_solve(mi355x_kkt_handle h, int nrhs, double* rhs, int ld)
{
    if (!h) return MI355X_KKT_FATAL;
    if (!h->factored || !h->numeric_ready) { h->err = "solve: no factorisation available"; return MI355X_KKT_FATAL; }
    if (h->sym.n == 0 || nrhs == 0) return MI355X_KKT_SUCCESS;
    try { if (!h->num->solve_host(nrhs, rhs, ld)) { h->err = h->num->error(); return MI355X_KKT_FATAL; } return MI355X_KKT_SUCCESS; }
    catch (...) { h->err = "solve: unexpected exception"; return MI355X_KKT_FATAL; }
}

int mi355x_kkt_solve_device(mi355x_kkt_handle h, int nrhs, double* drhs, int ld)
{
    if (!h) return MI355X_KKT_FATAL;
    if (!h->factored || !h->numeric_ready) { h->err = "solve: no factorisation available"; return MI355X_KKT_FATAL; }
    if (h->sym.n == 0 || nrhs == 0) return MI355X_KKT_SUCCESS;
    try { if (!h->num->solve_device(nrhs, drhs, ld)) { h->err = h->num->error(); return MI355X_KKT_FATAL; } return MI355X_KKT_SUCCESS; }
    catch (...) { h->err = "solve: unexpected exception"; return MI355X_KKT_FATAL; }
}

int mi355x_kkt_solve_device2(mi355x_kkt_handle h, int nrhs, const double* db, int ldb, double* dx, int ldx)
{
    if (!h) return MI355X_KKT_FATAL;
    if (!h->factored || !h->numeric_ready) { h->err = "solve: no factorisation available"; return MI355X_KKT_FATAL; }
    if (h->sym.n == 0 || nrhs == 0) return MI355X_KKT_SUCCESS;
    try { if (!h->num->solve_device2(nrhs, db, ldb, dx, ldx)) { h->err = h->num->error(); return MI355X_KKT_FATAL; } return MI355X_KKT_SUCCESS; }
    catch (...) { h->err = "solve: unexpected exception"; return MI355X_KKT_FATAL; }
}

int mi355x_kkt_set_pivtol(mi355x_kkt_handle h, double u)
{
    if (!h) return MI355X_KKT_FATAL;
    if (!(u > 0.0) || u > 0.5) { h->err = "set_pivtol: u must be in (0, 0.5]"; return MI355X_KKT_FATAL; }
    h->opts.pivtol = u;
    h->zero_delay_futile = false; h->zero_futile_events = 0;   // (not delays_exhausted: the growth cap is a property of the structure)
    if (h->num) h->num->set_pivtol(u);
    return MI355X_KKT_SUCCESS;
}

int mi355x_kkt_set_delay_rounds(mi355x_kkt_handle h, int rounds)
{
    if (!h || rounds < 0) return MI355X_KKT_FATAL;
    h->opts.delay_rounds = rounds;
    h->zero_delay_futile = false; h->zero_futile_events = 0; h->delays_exhausted = false;
    return MI355X_KKT_SUCCESS;
}

int mi355x_kkt_set_pivtolmax(mi355x_kkt_handle h, double umax)
{
    if (!h) return MI355X_KKT_FATAL;
    if (!(umax > 0.0) || umax > 0.5) { h->err = "set_pivtolmax: u must be in (0, 0.5]"; return MI355X_KKT_FATAL; }
    h->opts.pivtolmax = umax;
    if (h->num) h->num->set_pivtolmax(umax > h->opts.pivtol ? umax : h->opts.pivtol);
    return MI355X_KKT_SUCCESS;
}

/* IncreaseQuality (IpSparseSymLinearSolverInterface.hpp:220).  u <- min(umax, u^0.75) as the MA27/MA97/SPRAL adapters do
 * (IpMa97SolverInterface.cpp:822-854) -- but only when that can change the factorisation: the last factorisation recorded
 * whether ANY pivot decision would come out differently at u = pivtolmax (a pivot that passes the threshold tests at umax
 * passes them at every smaller u).  Returns 1 and stores the new u when the caller should refactor, 0 when the quality
 * cannot be increased (u already at its maximum, or no decision depends on u: Ipopt then goes straight to its
 * perturbation fallback instead of burning refactorisations, IpPDFullSpaceSolver.cpp:290-301).  That shortcut is OPT-IN
 * (opts.smart_quality, adapter option mi355x_smart_quality): by default u is raised whenever it is below its maximum, exactly as the
 * reference adapters do, so that Ipopt's 'q' info character and the refactorisation appear where they appear with MA27 / MA97. */
int mi355x_kkt_increase_quality(mi355x_kkt_handle h, double* new_u)
{
    if (!h) return 0;
    const double umax = h->opts.pivtolmax > h->opts.pivtol ? h->opts.pivtolmax : h->opts.pivtol;
    if (h->opts.pivtol >= umax) return 0;
    if (h->opts.smart_quality && h->factored && !h->stats_stale && !h->last.u_sensitive) return 0;      // (opt-in: the reference adapters always raise u)
    if (h->num && (h->opts.scaling == 4 || h->opts.scaling == 6)) h->num->invalidate_matching();      // a reused matching scaling is computed afresh when the caller asks for better quality
    double u = std::pow(h->opts.pivtol, 0.75);
    if (u > umax) u = umax;
    h->opts.pivtol = u;
    if (h->num) h->num->set_pivtol(u);
    if (new_u) *new_u = u;
    return 1;
}

int mi355x_kkt_get_info(mi355x_kkt_handle h, mi355x_kkt_info* info)
{
    if (!h || !info) return MI355X_KKT_FATAL;
    std::memset(info, 0, sizeof(*info));
    const Symbolic& S = h->sym;
    info->n = S.n; info->nnz_in = S.nnz_in; info->nnz_a = S.nnz_a; info->nnz_l = S.nnz_l;
    info->flops_factor = S.flops_factor; info->flops_solve = 4 * S.nnz_l - 3 * (int64_t)S.n;
    info->bytes_factor = 12 * (int64_t)S.nnz_a + 8 * S.nnz_l + 4 * S.sum_sn_rows;
    info->bytes_solve = 2 * (8 * S.nnz_l + 4 * S.sum_sn_rows) + 24 * (int64_t)S.n;
    info->sum_sn_rows = S.sum_sn_rows; info->cb_doubles = S.cb_doubles;
    info->num_sn = S.num_sn; info->num_levels = S.num_levels; info->maxfront = S.maxfront; info->maxsupernode = S.maxsupernode;
    info->num_pairs = S.num_pairs; info->num_big_fronts = S.num_big;
    info->num_neg = h->last.num_neg; info->num_zero = h->last.num_zero; info->num_two = h->last.num_two; info->num_small = h->last.num_small;
    info->u_sensitive = h->factored ? h->last.u_sensitive : 1; info->pivtol = h->opts.pivtol;
    info->num_fast_blocks = h->factored ? h->last.num_fast : 0;
    info->num_delayed = h->num_delayed; info->num_restructures = h->num_restructures;
    info->time_analyse = S.time_analyse;
    if (h->num) { info->time_factor_ms = h->num->last_factor_ms(); info->time_solve_ms = h->num->last_solve_ms(); h->num->matching_stats(&info->matching_ms, &info->matching_rounds, &info->matching_unmatched); }
    return MI355X_KKT_SUCCESS;
}

int mi355x_kkt_get_symbolic(mi355x_kkt_handle h, int what, int* out, int64_t cap)
{
    if (!h || !h->analysed || !out) return MI355X_KKT_FATAL;
    const Symbolic& S = h->sym;
    const avec<int>* v = nullptr;
    if (what == 27) {      // the storage plan of the contribution blocks (symbolic.cpp step 12a): {window (0: every block resident), doubles of all blocks if resident, doubles never reused} as (lo, hi) halves
        if (cap < 5) { h->err = "get_symbolic: buffer too small"; return MI355X_KKT_FATAL; }
        out[0] = S.cb_window;
        out[1] = (int)(S.cb_plain_doubles & 0xffffffffll); out[2] = (int)(S.cb_plain_doubles >> 32);
        out[3] = (int)(S.cb_resident_doubles & 0xffffffffll); out[4] = (int)(S.cb_resident_doubles >> 32);
        return MI355X_KKT_SUCCESS;
    }
    if (what == 28) {      // cb_off of every front as (low, high) 32-bit halves: 2 * num_sn ints (tests: no two blocks alive together share space)
        if (cap < 2 * (int64_t)S.num_sn) { h->err = "get_symbolic: buffer too small"; return MI355X_KKT_FATAL; }
        for (int s2 = 0; s2 < S.num_sn; ++s2) { out[2 * s2] = (int)(S.cb_off[s2] & 0xffffffffll); out[2 * s2 + 1] = (int)(S.cb_off[s2] >> 32); }
        return MI355X_KKT_SUCCESS;
    }
    switch (what) {
        case 0: v = &S.perm; break;        case 1: v = &S.sn_colptr; break;  case 2: v = &S.sn_rowptr; break;
        case 3: v = &S.sn_rows; break;     case 4: v = &S.sn_parent; break;  case 5: v = &S.sn_level; break;
        case 6: v = &S.rel; break;         case 7: v = &S.acolptr; break;    case 8: v = &S.arow; break;
        case 9: v = &S.trip2slot; break;   case 10: v = &S.pair_of; break;   case 11: v = &S.sn_owner; break;
        case 12: v = &S.apos; break;       case 13: v = &S.level_ptr; break; case 14: v = &S.level_sn; break;
        case 15: v = &S.grp_pos; break;    case 16: v = &S.grp_rem; break;   case 17: v = &S.alias_child; break;
        case 18: v = &S.sn_glo; break;     case 19: v = &S.sn_gsz; break;    case 20: v = &S.sn_gdepth; break;
        case 21: v = &S.dup_ptr; break;    case 22: v = &S.dup_src; break;    case 23: v = &S.sn_class; break;
        case 24: v = &S.rslot_ptr; break;  case 25: v = &S.rslot_idx; break;  case 26: v = &S.rslot_col; break;      // the symmetric row view (equilibration, refinement)
        default: h->err = "get_symbolic: unknown selector"; return MI355X_KKT_FATAL;
    }
    if ((int64_t)v->size() > cap) { h->err = "get_symbolic: buffer too small"; return MI355X_KKT_FATAL; }
    if (!v->empty()) std::memcpy(out, v->data(), v->size() * sizeof(int));
    return MI355X_KKT_SUCCESS;
}

int mi355x_kkt_profile(mi355x_kkt_handle h, int reps, double* ms, int* launches, int capacity)
{
    if (!h || !ms || !launches) return MI355X_KKT_FATAL;
    if (!h->numeric_ready || !h->factored) { h->err = "profile: factor() first"; return MI355X_KKT_FATAL; }
    if (capacity < MI355X_KKT_KERNEL_COUNT) { h->err = "profile: capacity too small"; return MI355X_KKT_FATAL; }
    try { if (!h->num->profile(reps > 0 ? reps : 1, ms, launches)) { h->err = h->num->error(); return MI355X_KKT_FATAL; } return MI355X_KKT_SUCCESS; }
    catch (...) { h->err = "profile: unexpected exception"; return MI355X_KKT_FATAL; }
}

/* development aid, not part of the public header: phase time stamps of one workgroup */
int mi355x_kkt_debug_clocks(mi355x_kkt_handle h, unsigned long long* out128)
{
    if (!h || !h->numeric_ready) return MI355X_KKT_FATAL;
    try { return h->num->debug_clocks(out128) ? 0 : MI355X_KKT_FATAL; } catch (...) { return MI355X_KKT_FATAL; }
}

/* development aid, not part of the public header: pivot data of the last factorisation (permuted numbering, pivot order inside a front) */
int mi355x_kkt_debug_pivots(mi355x_kkt_handle h, double* dinv, double* doff, int* ptype, int* lperm)
{
    if (!h || !h->numeric_ready || !h->factored) return MI355X_KKT_FATAL;
    try { return h->num->debug_pivots(dinv, doff, ptype, lperm) ? 0 : MI355X_KKT_FATAL; } catch (...) { return MI355X_KKT_FATAL; }
}

// ---- multi-GPU ----
int mi355x_kkt_comm_unique_id(void* out128)
{
    if (!out128) return MI355X_KKT_FATAL;
    try { std::string err; return Numeric::rccl_unique_id(out128, err) ? MI355X_KKT_SUCCESS : MI355X_KKT_FATAL; } catch (...) { return MI355X_KKT_FATAL; }
}
int mi355x_kkt_set_comm_rccl(mi355x_kkt_handle h, const void* unique_id128)
{
    if (!h || !unique_id128) return MI355X_KKT_FATAL;
    if (!h->numeric_ready) { h->err = std::string("set_comm_rccl: no device set-up") + (h->setup_err.empty() ? " (analyse() first, and a usable HIP device; no CPU fallback)" : ": " + h->setup_err); return MI355X_KKT_FATAL; }
    try { if (!h->num->set_comm_rccl(unique_id128)) { h->err = h->num->error(); return MI355X_KKT_FATAL; } return MI355X_KKT_SUCCESS; } catch (...) { return MI355X_KKT_FATAL; }
}
int mi355x_kkt_set_comm_callbacks(mi355x_kkt_handle h, mi355x_kkt_allreduce_fn fn, void* ctx)
{
    if (!h) return MI355X_KKT_FATAL;
    if (!h->numeric_ready) { h->err = std::string("set_comm_callbacks: no device set-up") + (h->setup_err.empty() ? " (analyse() first, and a usable HIP device; no CPU fallback)" : ": " + h->setup_err); return MI355X_KKT_FATAL; }
    try { if (!h->num->set_comm_callback(fn, ctx)) { h->err = h->num->error(); return MI355X_KKT_FATAL; } return MI355X_KKT_SUCCESS; } catch (...) { return MI355X_KKT_FATAL; }
}
/* the host-staged communicator over POSIX shared memory (comm_shm.cpp): ranks of ONE node, which may share a device */
int mi355x_kkt_comm_shm_id(void* out128, int nranks)
{
    if (!out128) return MI355X_KKT_FATAL;
    try { std::string err; if (!shm_comm_create(nranks, out128, err)) { fprintf(stderr, "[mi355x_kkt] %s\n", err.c_str()); return MI355X_KKT_FATAL; } return MI355X_KKT_SUCCESS; } catch (...) { return MI355X_KKT_FATAL; }
}
void mi355x_kkt_comm_shm_discard(const void* id128) { try { shm_comm_discard(id128); } catch (...) {} }
int mi355x_kkt_set_comm_shm(mi355x_kkt_handle h, const void* id128)
{
    if (!h || !id128) return MI355X_KKT_FATAL;
    if (!h->numeric_ready) { h->err = std::string("set_comm_shm: no device set-up") + (h->setup_err.empty() ? " (analyse() first, and a usable HIP device; no CPU fallback)" : ": " + h->setup_err); return MI355X_KKT_FATAL; }
    try {
        ShmComm* c = shm_comm_attach(id128, h->opts.rank, h->opts.nranks > 0 ? h->opts.nranks : 1, h->err);
        if (!c) return MI355X_KKT_FATAL;
        if (!h->num->set_comm_callback(shm_comm_allreduce, c) || !h->num->set_comm_range_callback(shm_comm_allreduce_range)) { h->err = h->num->error(); shm_comm_destroy(c); return MI355X_KKT_FATAL; }
        shm_comm_destroy(h->shm); h->shm = c;
        return MI355X_KKT_SUCCESS;
    } catch (...) { h->err = "set_comm_shm: unexpected exception"; return MI355X_KKT_FATAL; }
}
int mi355x_kkt_set_comm_range_callback(mi355x_kkt_handle h, mi355x_kkt_allreduce_range_fn fn)
{
    if (!h) return MI355X_KKT_FATAL;
    if (!h->numeric_ready) { h->err = std::string("set_comm_range_callback: no device set-up") + (h->setup_err.empty() ? " (analyse() first, and a usable HIP device; no CPU fallback)" : ": " + h->setup_err); return MI355X_KKT_FATAL; }
    try { return h->num->set_comm_range_callback(fn) ? MI355X_KKT_SUCCESS : MI355X_KKT_FATAL; } catch (...) { return MI355X_KKT_FATAL; }
}
/* host only: the collectives one rank issues (see include/mi355x_kkt.h) -- needs the analysis, not a device */
int mi355x_kkt_comm_plan(mi355x_kkt_handle h, int rank, int range_local, int* records6, int capacity_records, int* count)
{
    if (!h || !count) return MI355X_KKT_FATAL;
    if (!h->analysed) { h->err = "comm_plan: analyse() first"; return MI355X_KKT_FATAL; }
    try {
        const int P = h->so.nranks > 0 ? h->so.nranks : 1;
        if (rank < 0 || rank >= P) { h->err = "comm_plan: no such rank"; return MI355X_KKT_FATAL; }
        std::vector<int> out;
        Numeric::comm_plan(h->sym, P, rank, range_local != 0, out);
        *count = (int)(out.size() / 6);
        if (records6) for (int i = 0; i < *count && i < capacity_records; ++i) for (int j = 0; j < 6; ++j) records6[6 * i + j] = out[6 * (size_t)i + j];
        return MI355X_KKT_SUCCESS;
    } catch (...) { h->err = "comm_plan: unexpected exception"; return MI355X_KKT_FATAL; }
}
int mi355x_kkt_comm_info(mi355x_kkt_handle h, int* kind, int* ranks_seen, int* range_local, int* exchange_steps)
{
    if (!h) return MI355X_KKT_FATAL;
    if (!h->numeric_ready) { h->err = std::string("comm_info: no device set-up") + (h->setup_err.empty() ? " (analyse() first, and a usable HIP device; no CPU fallback)" : ": " + h->setup_err); return MI355X_KKT_FATAL; }
    try { h->num->comm_info(kind, ranks_seen, range_local, exchange_steps); return MI355X_KKT_SUCCESS; } catch (...) { return MI355X_KKT_FATAL; }
}
int mi355x_kkt_exchange_bytes(mi355x_kkt_handle h, int64_t* arena_bytes, int64_t* rhs_bytes)
{
    if (!h || !h->numeric_ready) return MI355X_KKT_FATAL;
    if (arena_bytes) *arena_bytes = h->num->exchange_bytes(0);
    if (rhs_bytes) *rhs_bytes = h->num->exchange_bytes(1);
    return MI355X_KKT_SUCCESS;
}
#define MG_GUARD if (!h) return MI355X_KKT_FATAL; if (!h->numeric_ready) { h->err = "multi-GPU call without a device"; return MI355X_KKT_FATAL; }
int mi355x_kkt_factor_local(mi355x_kkt_handle h, const double* dvals) { MG_GUARD try { if (!h->num->factor_local(dvals)) { h->err = h->num->error(); return MI355X_KKT_FATAL; } return 0; } catch (...) { return MI355X_KKT_FATAL; } }
int mi355x_kkt_top_arena(mi355x_kkt_handle h, double** d, int64_t* nd) { MG_GUARD try { if (!h->num->top_arena(d, nd)) { h->err = h->num->error(); return MI355X_KKT_FATAL; } return 0; } catch (...) { return MI355X_KKT_FATAL; } }
int mi355x_kkt_factor_top(mi355x_kkt_handle h, int* nneg, int* nzero)
{
    MG_GUARD
    try { FactorStats st; if (!h->num->factor_top(st)) { h->err = h->num->error(); return MI355X_KKT_FATAL; }
          h->last = st; h->factored = true; h->stats_stale = false; if (nneg) *nneg = st.num_neg; if (nzero) *nzero = st.num_zero; return 0; } catch (...) { return MI355X_KKT_FATAL; }
}
int mi355x_kkt_solve_fwd_local(mi355x_kkt_handle h, double* d) { MG_GUARD try { if (!h->num->solve_fwd_local(d)) { h->err = h->num->error(); return MI355X_KKT_FATAL; } return 0; } catch (...) { return MI355X_KKT_FATAL; } }
int mi355x_kkt_top_rhs(mi355x_kkt_handle h, double** d, int64_t* nd) { MG_GUARD try { if (!h->num->top_rhs(d, nd)) { h->err = h->num->error(); return MI355X_KKT_FATAL; } return 0; } catch (...) { return MI355X_KKT_FATAL; } }
int mi355x_kkt_solve_top_and_bwd(mi355x_kkt_handle h, double* d) { MG_GUARD try { if (!h->num->solve_top_and_bwd(d)) { h->err = h->num->error(); return MI355X_KKT_FATAL; } return 0; } catch (...) { return MI355X_KKT_FATAL; } }

} // extern "C"

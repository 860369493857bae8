// symbolic.cpp -- host symbolic analysis (see symbolic.h).  Pure C++17, no GPU.
//
// Pipeline:  canonical lower pattern + duplicate maps  ->  zero-diagonal 2x2 pre-pairing
// (bipartite matching)  ->  compressed graph  ->  nested dissection with minimum-degree
// leaves  ->  elimination tree, postorder (pairs kept adjacent)  ->  column counts
// (skeleton / least-common-ancestor method)  ->  fundamental + relaxed supernodes  ->
// supernodal row structures, child->parent relative indices, A->front scatter map,
// level schedule, storage offsets, multi-GPU subtree ownership.
#include "symbolic.h"
#include "env_knobs.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <map>
#include <set>
#include <cstdio>
#include <sys/mman.h>
#include <unistd.h>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <atomic>
#include <functional>

namespace mi355x {

// ---- BlockCache: one reserved range of address space; big blocks are carved out of it first-fit with coalescing, pages stay resident
//      while an analysis runs (that is the point: no page faults for the 2nd .. n-th array), free ranges are given back to the kernel
//      (MADV_DONTNEED) when the last running analysis returns.  Everything else -- small blocks, blocks when the range is exhausted or
//      could not be reserved -- is malloc.
namespace {
struct CacheState {
    std::mutex mu;
    int scopes = 0;
    char* base = nullptr; size_t size = 0;        // the reservation (lazily made, kept for the life of the process: address space only)
    size_t top = 0;                               // allocation frontier: everything in use or on the free list lies below it
    size_t hwm = 0;                               // highest frontier since pages were last given back: [top, hwm) is resident but unused
    std::map<size_t, size_t> free_at;             // offset -> length of the free ranges below top, coalesced
    std::map<size_t, size_t> live;                // offset -> length of the blocks in use
    bool tried = false;
};
CacheState& cache_state() { static CacheState* st = new CacheState; return *st; }      // (never destroyed: vectors may be freed during static destruction)
constexpr size_t CACHE_MIN_BYTES = 256u << 10, CACHE_ALIGN = 4096;
void cache_release_range(CacheState& C, size_t off, size_t len)      // physical pages of a free range back to the kernel (whole pages inside it)
{
    const size_t a = (off + CACHE_ALIGN - 1) & ~(CACHE_ALIGN - 1), b = (off + len) & ~(CACHE_ALIGN - 1);
    if (b > a) (void)madvise(C.base + a, b - a, MADV_DONTNEED);
}
}
void* BlockCache::get(size_t bytes)
{
    if (bytes >= CACHE_MIN_BYTES) {
        CacheState& C = cache_state();
        std::lock_guard<std::mutex> lk(C.mu);
        if (C.scopes > 0) {
            if (!C.tried) {
                C.tried = true;
                const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGESIZE);
                size_t want = (pages > 0 && psz > 0) ? (size_t)pages * (size_t)psz / 2 : ((size_t)16 << 30);
                want = std::min(want, (size_t)256 << 30);
                void* p = mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
                if (p != MAP_FAILED) { C.base = static_cast<char*>(p); C.size = want; }
            }
            if (C.base) {
                const size_t len = (bytes + CACHE_ALIGN - 1) & ~(CACHE_ALIGN - 1);
                // best fit among the free ranges (pages already resident); otherwise fresh space at the top
                auto best = C.free_at.end();
                for (auto it = C.free_at.begin(); it != C.free_at.end(); ++it)
                    if (it->second >= len && (best == C.free_at.end() || it->second < best->second)) best = it;
                if (best != C.free_at.end()) {
                    const size_t off = best->first, flen = best->second;
                    C.free_at.erase(best);
                    if (flen > len) C.free_at.emplace(off + len, flen - len);
                    C.live[off] = len;
                    return C.base + off;
                }
                if (C.top + len <= C.size) { const size_t off = C.top; C.top += len; C.hwm = std::max(C.hwm, C.top); C.live[off] = len; return C.base + off; }
            }
        }
    }
    void* p = std::malloc(bytes ? bytes : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void BlockCache::put(void* p, size_t bytes) noexcept
{
    if (!p) return;
    if (bytes < CACHE_MIN_BYTES) { std::free(p); return; }       // (the range never serves a request below the threshold)
    CacheState& C = cache_state();
    char* q = static_cast<char*>(p);
    std::lock_guard<std::mutex> lk(C.mu);
    if (!C.base || q < C.base || q >= C.base + C.size) { std::free(p); return; }
    size_t off = (size_t)(q - C.base);
    auto it = C.live.find(off);
    if (it == C.live.end()) return;               // (cannot happen)
    size_t len = it->second;
    C.live.erase(it);
    // coalesce with the free neighbours
    auto nx = C.free_at.lower_bound(off);
    if (nx != C.free_at.end() && nx->first == off + len) { len += nx->second; nx = C.free_at.erase(nx); }
    if (nx != C.free_at.begin()) { auto pv = std::prev(nx); if (pv->first + pv->second == off) { off = pv->first; len += pv->second; C.free_at.erase(pv); } }
    if (off + len == C.top) C.top = off;          // the frontier moves down (its pages stay resident while an analysis runs)
    else C.free_at.emplace(off, len);
    if (C.scopes == 0) {                          // a block that outlived its analysis (the Symbolic arrays): its pages go back at once
        if (C.top == off) { cache_release_range(C, C.top, C.hwm - C.top); C.hwm = C.top; }
        else cache_release_range(C, off, len);
    }
}
BlockCache::Scope::Scope() { CacheState& C = cache_state(); std::lock_guard<std::mutex> lk(C.mu); ++C.scopes; if (C.scopes == 1 && !C.tried && knob_disabled("blockcache")) C.tried = true; }      // (development knob: no reservation => every block is malloc)
BlockCache::Scope::~Scope()
{
    CacheState& C = cache_state();
    std::lock_guard<std::mutex> lk(C.mu);
    if (--C.scopes > 0 || !C.base) return;
    // nothing is being analysed any more: the free ranges (and what lies between the last block and the high-water mark) go back to the kernel.
    // (0.02 s after an analysis of LukVlE1 10^6, 0.08 s after synth_1e6 on the GPU box's host -- the difference between the analysis' own clock and
    // the caller's.  Measured and not kept, r05: the same from a detached thread -- the device set-up that follows, stream creation and pinned
    // allocations, then waits for the address-space lock the page release holds, 0.01 -> 0.07 s: nothing gained.)
    for (auto& e : C.free_at) cache_release_range(C, e.first, e.second);
    if (C.hwm > C.top) cache_release_range(C, C.top, C.hwm - C.top);
    C.hwm = C.top;
}

namespace {

template <class T> using vector = avec<T>;       // every array of the analysis comes from the recycling allocator (symbolic.h); small ones fall through to malloc


// static-chunk parallel loop on std::thread (the analysis is the only multi-threaded host code; T <= 16)
// default: up to 32 threads (measured on the 2 x 64-core host of the GPU box); MI355X_KKT_THREADS overrides
int analysis_threads()
{
    const unsigned h = std::thread::hardware_concurrency();
    int t = h ? std::min((int)h, 32) : 1;
    if (const char* e = getenv("MI355X_KKT_THREADS")) t = atoi(e);
    return std::max(1, std::min(t, 256));
}
// The parallel loops of one analysis share a set of parked worker threads (PoolScope, opened by analyse / restructure_delays on the thread that runs the
// analysis): a region costs a wake-up instead of creating and joining T threads -- ~0.8 ms per region at T = 32, some 40 regions per analysis.
struct WorkerPool {
    int nworkers;
    std::vector<std::thread> th;
    std::mutex mu; std::condition_variable cv, cvd;
    const std::function<void(int)>* job = nullptr;
    int ntasks = 0, remaining = 0; unsigned gen = 0; bool stop = false;
    explicit WorkerPool(int n) : nworkers(n) { th.reserve(n); for (int w = 0; w < n; ++w) th.emplace_back([this, w] { loop(w); }); }
    ~WorkerPool() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); for (auto& x : th) x.join(); }
    void loop(int w) {
        unsigned seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
            const std::function<void(int)>* j = job; const int nt = ntasks;
            lk.unlock();
            if (w + 1 < nt) (*j)(w + 1);
            lk.lock();
            if (--remaining == 0) cvd.notify_one();
        }
    }
    void run(int T, const std::function<void(int)>& fn) {      // fn(0) on the caller, fn(1 .. T-1) on the workers
        { std::lock_guard<std::mutex> lk(mu); job = &fn; ntasks = T; remaining = nworkers; ++gen; }
        cv.notify_all();
        fn(0);
        std::unique_lock<std::mutex> lk(mu);
        cvd.wait(lk, [&] { return remaining == 0; });
    }
};
thread_local WorkerPool* tl_pool = nullptr;
struct PoolScope {
    WorkerPool* mine = nullptr;
    explicit PoolScope(int T) { if (!tl_pool && T > 1 && !knob_disabled("thread_pool")) { mine = new WorkerPool(T - 1); tl_pool = mine; } }
    ~PoolScope() { if (mine) { tl_pool = nullptr; delete mine; } }
};
template <class F> void parallel_chunks(long long n, int T, F fn) {
    if (T <= 1 || n < 4096) { fn(0LL, n, 0); return; }
    if (tl_pool && T - 1 <= tl_pool->nworkers) {
        const std::function<void(int)> job = [&](int t) { fn(n * t / T, n * (t + 1) / T, t); };
        tl_pool->run(T, job);
        return;
    }
    std::vector<std::thread> th; th.reserve(T);
    for (int t = 0; t < T; ++t) { long long b = n * t / T, e = n * (t + 1) / T; th.emplace_back([=, &fn] { fn(b, e, t); }); }
    for (auto& x : th) x.join();
}

// Parallel bucketing of items 0..N-1 by key (0..nb-1): ptr[nb + 1] and the items of every bucket.  Counts and cursors are bumped with relaxed
// atomic adds, so the order INSIDE a bucket depends on the thread timing: every caller sorts its buckets afterwards (they need sorted buckets
// anyway), which makes the result deterministic again.
template <class KeyFn> void parallel_bucket(long long N, int nb, int T, KeyFn key, vector<int>& ptr, vector<int>& items)
{
    ptr.assign((size_t)nb + 1, 0);
    parallel_chunks(N, T, [&](long long b, long long e, int) { for (long long i = b; i < e; ++i) __atomic_fetch_add(&ptr[key(i) + 1], 1, __ATOMIC_RELAXED); });
    for (int j = 0; j < nb; ++j) ptr[j + 1] += ptr[j];
    items.resize((size_t)N);
    vector<int> pos(ptr.begin(), ptr.end() - 1);
    parallel_chunks(N, T, [&](long long b, long long e, int) { for (long long i = b; i < e; ++i) items[__atomic_fetch_add(&pos[key(i)], 1, __ATOMIC_RELAXED)] = (int)i; });
}

// ---------------------------------------------------------------------------------------
// 1. canonical lower pattern in the ORIGINAL numbering (col = min, row = max), dedup
// ---------------------------------------------------------------------------------------
struct Pattern {
    vector<int> colptr, row;   // lower CSC, rows sorted, diagonal present
    vector<int> t2slot;        // triplet -> slot
    vector<int> tcnt, tsorted; // triplets grouped by column (tcnt: n + 1 offsets), inside a column by (row, triplet index)
    vector<int> sfirst, scnt;  // per slot: where its triplets start in tsorted and how many there are (0: a diagonal nobody supplied) -- ascending triplet index
};

bool build_pattern(int n, int nnz, const int* ri, const int* ci, int base, int format, Pattern& P,
                   std::string& err)
{
    const int T = analysis_threads();
    vector<int> crow;                      // CSR input only: the row of every entry
    if (format == 0) {
        std::atomic<int> bad(0);
        parallel_chunks(nnz, T, [&](long long tb, long long te, int) {
            for (long long t = tb; t < te; ++t) {
                const int r = ri[t] - base, c = ci[t] - base;
                if (r < 0 || r >= n || c < 0 || c >= n) bad.store(1, std::memory_order_relaxed);
            }
        });
        if (bad.load()) { err = "analyse: index out of range"; return false; }
    } else {  // CSR upper: ri = ia[n+1], ci = ja[nnz]
        if (ri[n] - base != nnz) { err = "analyse: ia[n] does not match nnz"; return false; }
        crow.resize(nnz);
        for (int i = 0; i < n; ++i)
            for (int p = ri[i] - base; p < ri[i + 1] - base; ++p) {
                int c = ci[p] - base;
                if (c < 0 || c >= n) { err = "analyse: index out of range"; return false; }
                crow[p] = i;
            }
    }
    // (col = min, row = max) of an entry, computed where it is needed: two arrays of the size of the input less to fault in
    const int* rowp = format == 0 ? ri : crow.data();
    const int rbase = format == 0 ? base : 0;
    auto LO = [&](long long t) { return std::min(rowp[t] - rbase, ci[t] - base); };
    auto HI = [&](long long t) { return std::max(rowp[t] - rbase, ci[t] - base); };
    // bucket by column (lo), then sort rows inside each column
    vector<int> cnt, sorted_t;
    parallel_bucket(nnz, n, T, LO, cnt, sorted_t);
    // per column: sort the (row, triplet) pairs, count distinct rows (+ the always-present diagonal) -- columns are
    // independent, so both passes run on threads; the prefix sum in between is sequential
    vector<int> sorted_hi(nnz), ndist(n, 0);
    parallel_chunks(n, T, [&](long long jb, long long je, int) {
        vector<std::pair<int,int>> tmp;
        for (int j = (int)jb; j < (int)je; ++j) {
            tmp.clear();
            for (int p = cnt[j]; p < cnt[j + 1]; ++p) tmp.emplace_back(HI(sorted_t[p]), sorted_t[p]);
            if (tmp.size() > 1) std::sort(tmp.begin(), tmp.end());
            int last = j, d = 1;
            for (size_t q = 0; q < tmp.size(); ++q) { sorted_hi[cnt[j] + q] = tmp[q].first; sorted_t[cnt[j] + q] = tmp[q].second; if (tmp[q].first != last) { ++d; last = tmp[q].first; } }
            ndist[j] = d;
        }
    });
    P.colptr.assign(n + 1, 0);
    for (int j = 0; j < n; ++j) P.colptr[j + 1] = P.colptr[j] + ndist[j];
    P.row.assign(P.colptr[n], 0);
    P.t2slot.assign(nnz, -1);
    P.sfirst.assign(P.colptr[n], 0); P.scnt.assign(P.colptr[n], 0);
    parallel_chunks(n, T, [&](long long jb, long long je, int) {
        for (int j = (int)jb; j < (int)je; ++j) {
            int w = P.colptr[j];
            P.row[w] = j;                       // diagonal first, always present
            P.sfirst[w] = cnt[j];
            int last = j;
            for (int p = cnt[j]; p < cnt[j + 1]; ++p) {
                if (sorted_hi[p] != last) { P.row[++w] = sorted_hi[p]; last = sorted_hi[p]; P.sfirst[w] = p; }
                P.t2slot[sorted_t[p]] = w; ++P.scnt[w];
            }
        }
    });
    P.colptr[n] = (int)P.row.size();
    P.tcnt = std::move(cnt); P.tsorted = std::move(sorted_t);
    return true;
}

// symmetric adjacency (no diagonal) from a lower pattern
void build_adjacency(int n, const vector<int>& colptr, const vector<int>& row, vector<int>& xadj, vector<int>& adj)
{
    const int T = analysis_threads();
    xadj.assign(n + 1, 0);
    parallel_chunks(n, T, [&](long long jb, long long je, int) {
        for (int j = (int)jb; j < (int)je; ++j)
            for (int p = colptr[j]; p < colptr[j + 1]; ++p) { const int i = row[p]; if (i != j) { __atomic_fetch_add(&xadj[i + 1], 1, __ATOMIC_RELAXED); __atomic_fetch_add(&xadj[j + 1], 1, __ATOMIC_RELAXED); } }
    });
    for (int i = 0; i < n; ++i) xadj[i + 1] += xadj[i];
    adj.resize(xadj[n]);
    vector<int> pos(xadj.begin(), xadj.end() - 1);
    parallel_chunks(n, T, [&](long long jb, long long je, int) {
        for (int j = (int)jb; j < (int)je; ++j)
            for (int p = colptr[j]; p < colptr[j + 1]; ++p) {
                const int i = row[p];
                if (i != j) { adj[__atomic_fetch_add(&pos[i], 1, __ATOMIC_RELAXED)] = j; adj[__atomic_fetch_add(&pos[j], 1, __ATOMIC_RELAXED)] = i; }
            }
    });
    // (the fill order depends on the thread timing: sorted lists make the adjacency -- and every ordering built on it -- deterministic;
    //  the serial construction produced exactly this order: neighbours below the node in column order, then the node's own column)
    parallel_chunks(n, T, [&](long long ib, long long ie, int) { for (int i = (int)ib; i < (int)ie; ++i) std::sort(adj.begin() + xadj[i], adj.begin() + xadj[i + 1]); });
}

// ---------------------------------------------------------------------------------------
// 2. zero-diagonal pre-pairing: maximum-cardinality bipartite matching (greedy by weight,
//    then augmenting paths) between zero-diagonal rows and non-zero-diagonal neighbours.
//    Cf. the "matching" orderings the reference can request from MA97/SPRAL
//    (IpMa97SolverInterface.cpp:654-674, IpSpralSolverInterface.cpp:199-204).
// ---------------------------------------------------------------------------------------
void zero_diag_matching(int n, const Pattern& P, const vector<int>& xadj, const vector<int>& adj,
                        const double* vals, int nnz, vector<int>& pair_of, int& num_pairs, vector<char>& zrow)
{
    pair_of.assign(n, -1); num_pairs = 0; zrow.assign(n, 0);
    if (!vals) return;
    // summed slot values
    vector<double> sval(P.row.size(), 0.0);
    const int T0 = analysis_threads();
    if ((int)P.tcnt.size() == n + 1 && (int)P.tsorted.size() == nnz) {
        // column by column on threads: the triplets of a column only touch that column's slots, and inside a slot they come in ascending triplet
        // index -- the order of the serial loop, so the sums are bit for bit the same
        parallel_chunks(n, T0, [&](long long jb, long long je, int) {
            for (long long p = P.tcnt[jb]; p < P.tcnt[je]; ++p) { const int t = P.tsorted[p]; sval[P.t2slot[t]] += vals[t]; }
        });
    } else for (int t = 0; t < nnz; ++t) sval[P.t2slot[t]] += vals[t];
    // per-node edge weights |a_ij| aligned with the (sorted) adjacency lists, by binary search in the lower pattern; row maxima from them
    const int T = analysis_threads();
    vector<double> diag(n, 0.0), rowmax(n, 0.0), w(adj.size(), 0.0);
    parallel_chunks(n, T, [&](long long ib, long long ie, int) {
        for (int i = (int)ib; i < (int)ie; ++i) {
            diag[i] = sval[P.colptr[i]];                                   // (the diagonal is the first entry of its column)
            double mx = 0.0;
            for (int p = xadj[i]; p < xadj[i + 1]; ++p) {
                const int v = adj[p], c = std::min(i, v), r = std::max(i, v);
                const int* b = P.row.data() + P.colptr[c]; const int* e = P.row.data() + P.colptr[c + 1];
                const int q = (int)(std::lower_bound(b, e, r) - P.row.data());
                const double a = std::fabs(sval[q]);
                w[p] = a; mx = std::max(mx, a);
            }
            rowmax[i] = mx;
        }
    });
    vector<char> isz(n, 0);
    int nz = 0;
    for (int i = 0; i < n; ++i)
        if (rowmax[i] > 0 && std::fabs(diag[i]) <= 1e-6 * rowmax[i]) { isz[i] = 1; ++nz; }
    if (nz == 0) return;
    vector<int> match_v(n, -1);   // for non-zero-diag node v: matched zero row
    vector<int> match_z(n, -1);   // for zero row z: matched v
    // greedy: zero rows in index order, best available neighbour by weight
    for (int z = 0; z < n; ++z) if (isz[z]) {
        int best = -1; double bw = 0;
        for (int p = xadj[z]; p < xadj[z + 1]; ++p) {
            int v = adj[p];
            if (isz[v] || match_v[v] >= 0 || w[p] <= 0) continue;
            if (w[p] > bw || (w[p] == bw && best >= 0 && v < best)) { bw = w[p]; best = v; }
        }
        if (best >= 0) { match_v[best] = z; match_z[z] = best; }
    }
    // augmenting paths (iterative DFS) for the rest
    vector<int> visit(n, -1), stack_z, stack_p, via(n, -1);
    for (int z0 = 0; z0 < n; ++z0) if (isz[z0] && match_z[z0] < 0) {
        stack_z.assign(1, z0); stack_p.assign(1, xadj[z0]);
        bool found = false; int vend = -1;
        while (!stack_z.empty() && !found) {
            int z = stack_z.back(); int& p = stack_p.back();
            if (p >= xadj[z + 1]) { stack_z.pop_back(); stack_p.pop_back(); continue; }
            int v = adj[p]; double wt = w[p]; ++p;
            if (isz[v] || wt <= 0 || visit[v] == z0) continue;
            visit[v] = z0; via[v] = z;
            if (match_v[v] < 0) { found = true; vend = v; }
            else { int z2 = match_v[v]; stack_z.push_back(z2); stack_p.push_back(xadj[z2]); }
        }
        if (found) {  // flip along the path
            int v = vend;
            while (v >= 0) { int z = via[v]; int vprev = match_z[z]; match_v[v] = z; match_z[z] = v; v = vprev; }
        }
    }
    for (int z = 0; z < n; ++z) if (isz[z] && match_z[z] >= 0) { pair_of[z] = match_z[z]; pair_of[match_z[z]] = z; zrow[z] = 1; ++num_pairs; }
    // second pass: small-diagonal rows that found no well-conditioned partner pair up among themselves
    // ([[0,a],[a,0]] is a perfectly good 2x2 pivot)
    for (int z = 0; z < n; ++z) if (isz[z] && pair_of[z] < 0) {
        int best = -1; double bw = 0;
        for (int p = xadj[z]; p < xadj[z + 1]; ++p) {
            int v = adj[p];
            if (!isz[v] || pair_of[v] >= 0 || w[p] <= 0) continue;
            if (w[p] > bw || (w[p] == bw && best >= 0 && v < best)) { bw = w[p]; best = v; }
        }
        if (best >= 0) { pair_of[z] = best; pair_of[best] = z; zrow[std::max(z, best)] = 1; ++num_pairs; }
    }
}

// ---------------------------------------------------------------------------------------
// 3. ordering on a (compressed) graph: nested dissection by level structures with
//    minimum-degree leaves
// ---------------------------------------------------------------------------------------
struct Graph { int n; vector<int> xadj, adj; };

// exact minimum degree with a quotient graph on the subgraph induced by `nodes`
// (neighbours outside the set are ignored).  Appends the elimination order to out.
class MinDegree {
public:
    explicit MinDegree(const Graph& g) : G(g), loc(g.n, -1), mark(g.n, -1) {}
    void order(const vector<int>& nodes, int* out) {
        int m = (int)nodes.size();
        if (m == 0) return;
        if (m == 1) { out[0] = nodes[0]; return; }
        for (int i = 0; i < m; ++i) loc[nodes[i]] = i;
        adjv.assign(m, {}); adje.assign(m, {}); members.assign(m, {});
        deg.assign(m, 0); alive.assign(m, 1); emark.assign(m, -1);
        for (int i = 0; i < m; ++i) {
            int g = nodes[i];
            for (int p = G.xadj[g]; p < G.xadj[g + 1]; ++p) { int l = loc[G.adj[p]]; if (l >= 0) adjv[i].push_back(l); }
            deg[i] = (int)adjv[i].size();
        }
        std::set<std::pair<int,int>> pq;
        for (int i = 0; i < m; ++i) pq.insert({deg[i], i});
        lmark.assign(m, -1); int stamp = 0;
        vector<int> Lp;
        for (int step = 0; step < m; ++step) {
            int p = pq.begin()->second; pq.erase(pq.begin());
            out[step] = nodes[p]; alive[p] = 0;
            // reach set of p
            Lp.clear(); ++stamp; lmark[p] = stamp;
            for (int v : adjv[p]) if (alive[v] && lmark[v] != stamp) { lmark[v] = stamp; Lp.push_back(v); }
            for (int e : adje[p]) for (int v : members[e]) if (alive[v] && lmark[v] != stamp) { lmark[v] = stamp; Lp.push_back(v); }
            // absorbed elements
            int estamp = ++stamp;
            for (int e : adje[p]) { emark[e] = estamp; members[e].clear(); members[e].shrink_to_fit(); }
            members[p] = Lp;
            int lpstamp = ++stamp;
            for (int v : Lp) lmark[v] = lpstamp;
            for (int i : Lp) {
                // prune adjv[i]: drop p, dead nodes and members of Lp (now reachable through element p)
                auto& av = adjv[i]; size_t k = 0;
                for (int v : av) if (alive[v] && lmark[v] != lpstamp) av[k++] = v;
                av.resize(k);
                auto& ae = adje[i]; k = 0;
                for (int e : ae) if (emark[e] != estamp) ae[k++] = e;
                ae.resize(k); ae.push_back(p);
            }
            // recompute exact degrees of the reach set
            for (int i : Lp) {
                int s = ++stamp; lmark[i] = s; int d = 0;
                for (int v : adjv[i]) if (lmark[v] != s) { lmark[v] = s; ++d; }
                for (int e : adje[i]) for (int v : members[e]) if (alive[v] && lmark[v] != s) { lmark[v] = s; ++d; }
                if (d != deg[i]) { pq.erase({deg[i], i}); deg[i] = d; pq.insert({d, i}); }
            }
            // restore Lp marks are stale now (stamps moved on) -- fine, every use re-stamps
        }
        for (int i = 0; i < m; ++i) loc[nodes[i]] = -1;
    }
private:
    const Graph& G;
    vector<int> loc, mark;
    vector<vector<int>> adjv, adje, members;
    vector<int> deg, lmark, emark; vector<char> alive;
};

class NestedDissection {
public:
    NestedDissection(const Graph& g, int leaf) : G(g), leaf_(std::max(leaf, 8)), md(g), tag(g.n, -1), lev(g.n, -1) {}
    MinDegree& mindeg() { return md; }
    struct Task { vector<int> nodes; int start; };
    // splits (or orders) every task on the stack until it is empty; with `budget` > 0 stops as soon as the stack holds
    // that many pending tasks (they are then handed to worker threads, each with its own NestedDissection state)
    void drain(vector<Task>& st, vector<int>& order) {
        while (!st.empty()) { Task t = std::move(st.back()); st.pop_back(); step(std::move(t), st, order); }
    }
    int leaf() const { return leaf_; }
    // one task: order it (leaf / unsplittable) or split it into sub-tasks pushed onto st
    void step(Task t, vector<Task>& st, vector<int>& order) {
        {
            int m = (int)t.nodes.size();
            if (m == 0) return;
            if (m <= leaf_) { md.order(t.nodes, order.data() + t.start); return; }
            // connected components of the induced subgraph
            ++stamp; for (int v : t.nodes) tag[v] = stamp;
            int cstamp = ++stamp;   // visited marker
            int pos = t.start; small.clear();
            vector<vector<int>> comps;
            for (int s : t.nodes) if (tag[s] == stamp - 1) {
                vector<int> comp; comp.push_back(s); tag[s] = cstamp;
                for (size_t q = 0; q < comp.size(); ++q) { int v = comp[q];
                    for (int p = G.xadj[v]; p < G.xadj[v + 1]; ++p) { int u = G.adj[p]; if (tag[u] == stamp - 1) { tag[u] = cstamp; comp.push_back(u); } } }
                comps.push_back(std::move(comp));
            }
            if (comps.size() > 1) {
                // big components become tasks; small ones are batched into leaf-sized MD calls
                for (auto& c : comps) {
                    if ((int)c.size() > leaf_) { int sz = (int)c.size(); st.push_back({std::move(c), pos}); pos += sz; }
                    else {
                        if ((int)(small.size() + c.size()) > leaf_ && !small.empty()) { md.order(small, order.data() + pos); pos += (int)small.size(); small.clear(); }
                        small.insert(small.end(), c.begin(), c.end());
                    }
                }
                if (!small.empty()) { md.order(small, order.data() + pos); pos += (int)small.size(); small.clear(); }
                return;
            }
            // one connected component: level structure from a pseudo-peripheral node
            vector<int>& comp = comps[0];
            int root = comp[0]; int nlev = 0;
            bool fresh = false;                                  // queue / lev hold the level structure of `root`
            for (int it = 0; it < 4; ++it) {
                int nl = bfs_levels(root, cstamp, ++stamp, queue); cstamp = stamp;
                fresh = true;
                // candidate: min-degree node of last level
                int best = -1, bd = 1 << 30;
                for (int q = (int)queue.size() - 1; q >= 0 && lev[queue[q]] == nl - 1; --q) {
                    int v = queue[q], d = G.xadj[v + 1] - G.xadj[v]; if (d < bd) { bd = d; best = v; } }
                if (nl <= nlev) { nlev = nl; break; }
                nlev = nl; if (best == root) break; root = best; fresh = false;
            }
            if (!fresh) { nlev = bfs_levels(root, cstamp, ++stamp, queue); cstamp = stamp; }      // (otherwise the pass just made IS the structure of root)
            if (nlev < 3) { md.order(t.nodes, order.data() + t.start); return; }
            // choose the separator level: small and balanced
            int bestsz = 0;                                  // size of the chosen level
            auto best_level = [&](int nl, int& bestl, double& bests) {
                vector<int> lsize(nl, 0);
                for (int v : queue) lsize[lev[v]]++;
                bestl = -1; bests = 1e300; int below = lsize[0];
                for (int l = 1; l <= nl - 2; ++l) {
                    int a = below, s = lsize[l], b = m - a - s; below += s;
                    if (a == 0 || b == 0) break;
                    double imb = std::fabs((double)a - b) / (double)(a + b);
                    double score = (double)s * (1.0 + 4.0 * imb * imb) + 0.05 * m * imb;
                    if (score < bests) { bests = score; bestl = l; bestsz = s; }
                }
            };
            int bestl = -1; double bests = 1e300;
            best_level(nlev, bestl, bests);
            // a separator of a handful of nodes that already halves the piece (banded / chain-like graphs: the LukVl family) cannot be beaten by
            // the two alternative level structures below: skip their breadth-first passes (3 of the ~8 per bisection)
            const bool tiny_sep = bestl >= 0 && bestsz <= std::max(4, m / 2000) && bests <= 1.2 * bestsz;
            // Second level structure, rooted at the whole LAST LEVEL of the first one.  On stencil-like graphs whose BFS balls
            // are boxes (9-point / block couplings, i.e. every PDE-constrained KKT) the levels from a corner are L-shaped, but
            // the last level is an entire side of the domain, and the levels grown from a side are straight lines: separators
            // up to 2.7x smaller at the top of the tree.  Keep whichever structure has the better separator.
            if (!tiny_sep) {
                vector<int> src;
                for (int q = (int)queue.size() - 1; q >= 0 && lev[queue[q]] == nlev - 1; --q) src.push_back(queue[q]);
                // second candidate source: HALF of that last level (on a square domain the last level is two sides meeting in
                // the far corner, and only one side gives straight levels): the half reached first by a BFS inside the level
                // from its lowest-degree vertex
                vector<int> half;
                if (src.size() >= 8) {
                    const int mark = ++stamp;                          // temporary tag for "in src, not visited"
                    int start = src[0], bd = 1 << 30;
                    for (int v : src) { tag[v] = mark; const int d = G.xadj[v + 1] - G.xadj[v]; if (d < bd) { bd = d; start = v; } }
                    half.push_back(start); tag[start] = cstamp;
                    for (size_t q = 0; q < half.size() && half.size() < src.size() / 2; ++q) {
                        const int v = half[q];
                        for (int p = G.xadj[v]; p < G.xadj[v + 1] && half.size() < src.size() / 2; ++p) { const int u = G.adj[p]; if (tag[u] == mark) { tag[u] = cstamp; half.push_back(u); } }
                    }
                    for (int v : src) tag[v] = cstamp;                 // restore
                    if (half.size() < src.size() / 4) half.clear();     // the level is not connected enough for this to mean anything
                }
                vector<int> queue0 = queue, lev0(queue.size());
                for (size_t q = 0; q < queue.size(); ++q) lev0[q] = lev[queue[q]];
                vector<int> queueb = queue0, levb = lev0;               // best structure so far
                for (int cand = 0; cand < 2; ++cand) {
                    const vector<int>& from = cand == 0 ? src : half;
                    if (from.empty()) continue;
                    const int nl2 = bfs_levels_multi(from, cstamp, ++stamp, queue); cstamp = stamp;
                    int bl2 = -1; double bs2 = 1e300;
                    if (nl2 >= 3 && (int)queue.size() == m) best_level(nl2, bl2, bs2);
                    if (bl2 >= 0 && bs2 < bests) {
                        bestl = bl2; bests = bs2; nlev = nl2; queueb = queue;
                        levb.resize(queue.size()); for (size_t q = 0; q < queue.size(); ++q) levb[q] = lev[queue[q]];
                    }
                }
                queue = queueb; for (size_t q = 0; q < queue.size(); ++q) lev[queue[q]] = levb[q];
                for (int v : queue) tag[v] = cstamp;                    // (all tags of the component end on the current stamp)
            }
            if (bestl < 0) { md.order(t.nodes, order.data() + t.start); return; }
            vector<int> A, B, S;
            for (int v : queue) {
                int l = lev[v];
                if (l < bestl) A.push_back(v);
                else if (l > bestl) B.push_back(v);
                else {   // thin the separator: a node with no neighbour in level bestl+1 joins A
                    bool touchesB = false;
                    for (int p = G.xadj[v]; p < G.xadj[v + 1]; ++p) { int u = G.adj[p]; if (tag[u] == cstamp && lev[u] == bestl + 1) { touchesB = true; break; } }
                    if (touchesB) S.push_back(v); else A.push_back(v);
                }
            }
            if (A.empty() || B.empty() || (int)S.size() * 2 > m) { md.order(t.nodes, order.data() + t.start); return; }
            int sa = (int)A.size(), sb = (int)B.size();
            // separator last; inside S keep BFS order (dense clique anyway)
            for (size_t i = 0; i < S.size(); ++i) order[t.start + sa + sb + (int)i] = S[i];
            st.push_back({std::move(A), t.start});
            st.push_back({std::move(B), t.start + sa});
        }
    }
    int stamp = 0;
    vector<int> queue, small;
private:
    // BFS restricted to nodes with tag == in_stamp; re-tags visited with out_stamp; returns #levels
    int bfs_levels(int root, int in_stamp, int out_stamp, vector<int>& queue) {
        queue.clear(); queue.push_back(root); tag[root] = out_stamp; lev[root] = 0; int nl = 1;
        for (size_t q = 0; q < queue.size(); ++q) { int v = queue[q];
            for (int p = G.xadj[v]; p < G.xadj[v + 1]; ++p) { int u = G.adj[p];
                if (tag[u] == in_stamp) { tag[u] = out_stamp; lev[u] = lev[v] + 1; nl = std::max(nl, lev[u] + 1); queue.push_back(u); } } }
        return nl;
    }
    int bfs_levels_multi(const vector<int>& roots, int in_stamp, int out_stamp, vector<int>& queue) {
        queue.clear(); int nl = 1;
        for (int r : roots) if (tag[r] == in_stamp) { tag[r] = out_stamp; lev[r] = 0; queue.push_back(r); }
        for (size_t q = 0; q < queue.size(); ++q) { int v = queue[q];
            for (int p = G.xadj[v]; p < G.xadj[v + 1]; ++p) { int u = G.adj[p];
                if (tag[u] == in_stamp) { tag[u] = out_stamp; lev[u] = lev[v] + 1; nl = std::max(nl, lev[u] + 1); queue.push_back(u); } } }
        return nl;
    }
    const Graph& G; int leaf_; MinDegree md; vector<int> tag, lev;
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

} // namespace

// ---------------------------------------------------------------------------------------
// Everything that follows from (permutation, permuted pattern, supernode partition): row structures, relative indices,
// levels, classes, multi-GPU ownership, storage.  Shared by analyse() and restructure_delays().
// ---------------------------------------------------------------------------------------
static bool finish_analysis(Symbolic& S, const SymbolicOptions& opt, const std::function<void(const char*)>& lap)
{
    const int n = S.n;
    const int nsn = S.num_sn;
    S.sn_of.resize(n);
    for (int s = 0; s < nsn; ++s) for (int j = S.sn_colptr[s]; j < S.sn_colptr[s + 1]; ++j) S.sn_of[j] = s;

    lap("supernodes");
    // ---- 9. supernodal row structures ----
    S.sn_rowptr.assign(nsn + 1, 0); S.sn_parent.assign(nsn, -1);
    S.sn_rows.clear(); S.sn_rows.reserve((size_t)n * 4);
    {
        vector<int> mark(n, -1), upd;
        vector<int> chead(nsn, -1), cnext(nsn, -1);
        // The update rows of a supernode = (rows of its A columns) U (update rows of its children), beyond its own columns, sorted.  The child with the LONGEST list
        // brings it sorted already: only what the others add is collected and sorted, then merged in -- on the 3-D family (fronts of 10^4 rows handed up a chain
        // link by link) the sort of every whole list was the largest phase of the analysis (MBndryCntrl_3D 78: 0.24 of 0.73 s on the GPU box's host).
        vector<int> merged;
        for (int s = 0; s < nsn; ++s) {
            int c0 = S.sn_colptr[s], c1 = S.sn_colptr[s + 1];
            upd.clear();
            int bigc = -1, bign = 0;
            for (int c = chead[s]; c != -1; c = cnext[c]) { const int nc = S.sn_rowptr[c + 1] - S.sn_rowptr[c] - (S.sn_colptr[c + 1] - S.sn_colptr[c]); if (nc > bign) { bign = nc; bigc = c; } }
            const int* B = nullptr; int nb = 0;
            if (bign >= 64) {
                const int kc = S.sn_colptr[bigc + 1] - S.sn_colptr[bigc];
                const int* b0 = S.sn_rows.data() + S.sn_rowptr[bigc] + kc; const int* b1 = S.sn_rows.data() + S.sn_rowptr[bigc + 1];
                B = std::lower_bound(b0, b1, c1); nb = (int)(b1 - B);
                for (int q = 0; q < nb; ++q) mark[B[q]] = s;
            } else bigc = -1;
            for (int j = c0; j < c1; ++j)
                for (int p = S.acolptr[j]; p < S.acolptr[j + 1]; ++p) { int i = S.arow[p]; if (i >= c1 && mark[i] != s) { mark[i] = s; upd.push_back(i); } }
            for (int c = chead[s]; c != -1; c = cnext[c]) {
                if (c == bigc) continue;
                int kc = S.sn_colptr[c + 1] - S.sn_colptr[c];
                for (int p = S.sn_rowptr[c] + kc; p < S.sn_rowptr[c + 1]; ++p) { int i = S.sn_rows[p]; if (i >= c1 && mark[i] != s) { mark[i] = s; upd.push_back(i); } }
            }
            std::sort(upd.begin(), upd.end());
            if (nb > 0) {      // (B points into sn_rows, which grows below: merge into a buffer first)
                merged.resize((size_t)nb + upd.size());
                std::merge(B, B + nb, upd.begin(), upd.end(), merged.begin());
                upd.swap(merged);
            }
            S.sn_rowptr[s] = (int)S.sn_rows.size();
            for (int j = c0; j < c1; ++j) S.sn_rows.push_back(j);
            S.sn_rows.insert(S.sn_rows.end(), upd.begin(), upd.end());
            S.sn_rowptr[s + 1] = (int)S.sn_rows.size();
            if (!upd.empty()) { int p = S.sn_of[upd[0]]; S.sn_parent[s] = p; cnext[s] = chead[p]; chead[p] = s; }
        }
    }
    S.sum_sn_rows = (int64_t)S.sn_rows.size();
    // ---- 9b. the side children of in-place chain links go DOWN the chain ----
    // A separator of a 3-D problem is a chain of 64-column supernodes; link j + 1 is factored IN PLACE in the contribution block of link j (its front IS
    // link j's update rows) and up to four consecutive links are one launch with ONE rank-256 update (step 12 below) -- if the links after the first have
    // no other child.  On the MBndryCntrl_3D family 2 of 3 chain links do have one: a dangling subtree whose contribution block is a handful of rows
    // (median 2), finished many levels below.  Its rows lie in the link's front, hence in the update part of EVERY link below it in the chain: the block
    // can be extend-added into any of those instead and rides up inside the chain's contribution blocks (the assembly tree may hang a child on any front
    // down the chain that holds its rows; the elimination order is untouched).  So such a child is handed to the LOWEST link of the chain that still lies
    // above it in the level order (no level changes: the child is lower than its new parent was already) -- normally the chain's first link, which is
    // assembled from its children anyway.  The links above come out pure: the chain groups form (N = 50: 212 of 285 GFlop of trailing updates at rank
    // 193-256, 31 before), their k_big_assemble launches disappear, and a group-end update may be split for the look-ahead (which needs the NEXT group
    // pure: handing the children to the first link of their own GROUP instead was measured -- no split anywhere, 36.9 ms against 32.7).
    // Sums are formed in a different order than without the pass.
    S.num_rehung = 0;
    if (opt.chain_purify && opt.nranks <= 1) {      // (one GPU only: a rehung child may carry a LARGER index than its new parent, which the subtree-to-rank mapping of step 11 does not expect)
        vector<int> head(nsn, -1), nxt(nsn, -1), height(nsn, 0), below(nsn, -1);
        for (int s = nsn - 1; s >= 0; --s) { const int p = S.sn_parent[s]; if (p >= 0) { nxt[s] = head[p]; head[p] = s; } }      // natural children, ascending
        for (int s = 0; s < nsn; ++s) {
            const int ms = S.sn_rowptr[s + 1] - S.sn_rowptr[s];
            int ac = -1;
            if (ms > 128)
                for (int c = head[s]; c != -1; c = nxt[c]) {
                    const int mc = S.sn_rowptr[c + 1] - S.sn_rowptr[c], kc = S.sn_colptr[c + 1] - S.sn_colptr[c];
                    if (mc > 128 && mc - kc == ms) { ac = c; break; }      // (the structural part of the alias condition of step 12)
                }
            below[s] = ac;
            int h = 0;
            for (int c = head[s]; c != -1; c = nxt[c]) {
                if (ac >= 0 && c != ac) {
                    int b = s, x = ac;
                    while (x >= 0 && height[x] > height[c]) { b = x; x = below[x]; }
                    if (b != s) { S.sn_parent[c] = b; ++S.num_rehung; continue; }      // (height[b] > height[c]: the level of b stays what it is)
                }
                h = std::max(h, height[c] + 1);
            }
            height[s] = h;
        }
    }
    // children lists (ascending)
    S.child_ptr.assign(nsn + 1, 0);
    for (int s = 0; s < nsn; ++s) if (S.sn_parent[s] >= 0) S.child_ptr[S.sn_parent[s] + 1]++;
    for (int s = 0; s < nsn; ++s) S.child_ptr[s + 1] += S.child_ptr[s];
    S.child_idx.resize(S.child_ptr[nsn]);
    { vector<int> pos(S.child_ptr.begin(), S.child_ptr.end() - 1);
      for (int s = 0; s < nsn; ++s) if (S.sn_parent[s] >= 0) S.child_idx[pos[S.sn_parent[s]]++] = s; }

    lap("row structures");
    {   // symmetric row view of the permuted pattern (equilibration gather)
        const int nnzA = S.nnz_a;
        S.rslot_ptr.assign(n + 1, 0);
        for (int q = 0; q < nnzA; ++q) { S.rslot_ptr[S.arow[q] + 1]++; if (S.arow[q] != S.acol[q]) S.rslot_ptr[S.acol[q] + 1]++; }
        for (int i = 0; i < n; ++i) S.rslot_ptr[i + 1] += S.rslot_ptr[i];
        S.rslot_idx.resize(S.rslot_ptr[n]);
        vector<int> pos(S.rslot_ptr.begin(), S.rslot_ptr.end() - 1);
        S.rslot_col.resize(S.rslot_ptr[n]);
        for (int q = 0; q < nnzA; ++q) {
            const int r = S.arow[q], c = S.acol[q];
            S.rslot_col[pos[r]] = c; S.rslot_idx[pos[r]++] = q;
            if (r != c) { S.rslot_col[pos[c]] = r; S.rslot_idx[pos[c]++] = q; }
        }
    }
    // ---- 10. relative indices, A scatter positions, levels, offsets, stats ----
    S.rel.assign(S.sn_rows.size(), -1);
    {
        const int T = analysis_threads();
        std::atomic<int> bad(0);
        parallel_chunks(nsn, T, [&](long long sb, long long se, int) {        // (supernodes are independent; chunks of consecutive supernodes carry similar work)
            for (int s = (int)sb; s < (int)se; ++s) {
                int p = S.sn_parent[s]; if (p < 0) continue;
                int k = S.sn_colptr[s + 1] - S.sn_colptr[s];
                int p0 = S.sn_colptr[p], p1 = S.sn_colptr[p + 1], kp = p1 - p0;
                int q = S.sn_rowptr[p] + kp, qe = S.sn_rowptr[p + 1];
                for (int t = S.sn_rowptr[s] + k; t < S.sn_rowptr[s + 1]; ++t) {
                    int r = S.sn_rows[t];
                    if (r < p1) { S.rel[t] = r - p0; continue; }
                    while (q < qe && S.sn_rows[q] < r) ++q;
                    if (q >= qe || S.sn_rows[q] != r) { bad.store(1, std::memory_order_relaxed); break; }
                    S.rel[t] = kp + (q - (S.sn_rowptr[p] + kp));
                }
            }
        });
        if (bad.load()) { S.error = "analyse: internal error (child row missing in parent front)"; return false; }
        S.apos.resize(S.nnz_a);
        parallel_chunks(nsn, T, [&](long long sb, long long se, int) {
            for (int s = (int)sb; s < (int)se; ++s) {
                int c0 = S.sn_colptr[s], c1 = S.sn_colptr[s + 1], k = c1 - c0;
                int m = S.sn_rowptr[s + 1] - S.sn_rowptr[s];
                for (int j = c0; j < c1; ++j) {
                    int q = S.sn_rowptr[s] + k, qe = S.sn_rowptr[s + 1];
                    for (int p = S.acolptr[j]; p < S.acolptr[j + 1]; ++p) {
                        int i = S.arow[p], lr;
                        if (i < c1) lr = i - c0;
                        else { while (q < qe && S.sn_rows[q] < i) ++q;
                               if (q >= qe || S.sn_rows[q] != i) { bad.store(2, std::memory_order_relaxed); break; }
                               lr = k + (q - (S.sn_rowptr[s] + k)); }
                        S.apos[p] = lr + (j - c0) * m;
                    }
                }
            }
        });
        if (bad.load()) { S.error = "analyse: internal error (A row missing in front)"; return false; }
    }
    S.sn_level.assign(nsn, 0);
    for (int s = 0; s < nsn; ++s) { int p = S.sn_parent[s]; if (p >= 0) S.sn_level[p] = std::max(S.sn_level[p], S.sn_level[s] + 1); }
    S.num_levels = 0; for (int s = 0; s < nsn; ++s) S.num_levels = std::max(S.num_levels, S.sn_level[s] + 1);
    S.sn_class.resize(nsn);
    for (int s = 0; s < nsn; ++s) {
        int64_t k = S.sn_colptr[s + 1] - S.sn_colptr[s], m = S.sn_rowptr[s + 1] - S.sn_rowptr[s];
        S.sn_class[s] = m <= 32 ? FC_WAVE : (m <= 64 ? FC_LDS64 : (m <= 128 ? FC_LDS128 : FC_BIG));
        if (S.sn_class[s] == FC_BIG) S.num_big++;
        S.nnz_l += k * m - k * (k - 1) / 2;
        for (int64_t j = 0; j < k; ++j) { int64_t c = m - j; S.flops_factor += (c - 1) * (c + 2); }
        S.maxfront = std::max<int>(S.maxfront, (int)m); S.maxsupernode = std::max<int>(S.maxsupernode, (int)k);
    }
    S.minv_off.resize(nsn);
    { int64_t mo = 0; for (int s = 0; s < nsn; ++s) { int64_t k = S.sn_colptr[s + 1] - S.sn_colptr[s]; S.minv_off[s] = mo; mo += k * k; } S.minv_doubles = mo; }
    // level schedule buckets (level, class)
    S.level_ptr.assign((size_t)S.num_levels * FC_COUNT + 1, 0);
    for (int s = 0; s < nsn; ++s) S.level_ptr[(size_t)S.sn_level[s] * FC_COUNT + S.sn_class[s] + 1]++;
    for (size_t b = 0; b + 1 < S.level_ptr.size(); ++b) S.level_ptr[b + 1] += S.level_ptr[b];
    S.level_sn.resize(nsn);
    { vector<int> pos(S.level_ptr.begin(), S.level_ptr.end() - 1);
      for (int s = 0; s < nsn; ++s) S.level_sn[pos[(size_t)S.sn_level[s] * FC_COUNT + S.sn_class[s]]++] = s; }

    lap("rel/apos/levels");
    // ---- 11. multi-GPU ownership: proportional subtree-to-rank mapping ----
    S.sn_owner.assign(nsn, opt.nranks > 1 ? -1 : 0);
    S.sn_glo.assign(nsn, 0); S.sn_gsz.assign(nsn, std::max(1, opt.nranks)); S.sn_gdepth.assign(nsn, 0); S.num_gdepths = 1;
    if (opt.nranks > 1 && opt.subcube) {
        // SUBTREE-TO-SUBCUBE mapping: a front of the top of the tree is needed only by the ranks whose subtrees lie beneath it, so it is
        // replicated on THAT range of ranks [glo, glo + gsz) instead of on all of them.  Recursive bisection of (set of sibling subtrees,
        // range of ranks): a set is split into two bins of about equal work (LPT), the range in proportion; the heaviest member of a set is
        // OPENED -- its front joins the replicated part of the range, its children join the set -- while that shortens the critical rank's
        // work (a lone member is always opened: that is how a separator chain stays with its range).  A range of one rank owns what is left.
        // gdepth = number of bisections above the range: the exchange steps run deepest ranges first, all ranges of one depth in ONE
        // collective (every rank takes part in every step, which is what keeps the sequence deadlock-free by construction).
        vector<double> work(nsn, 0.0), own(nsn, 0.0);
        for (int s = 0; s < nsn; ++s) {
            double k = S.sn_colptr[s + 1] - S.sn_colptr[s], m = S.sn_rowptr[s + 1] - S.sn_rowptr[s];
            own[s] = k * m * m + 64.0; work[s] += own[s];
            if (S.sn_parent[s] >= 0) work[S.sn_parent[s]] += work[s];
        }
        struct Split { vector<int> bin[2]; double load[2]; int g[2]; double crit; };
        auto split_set = [&](const vector<int>& set, int g) {
            vector<int> order(set);
            std::sort(order.begin(), order.end(), [&](int a, int b) { return work[a] != work[b] ? work[a] > work[b] : a < b; });
            Split best; best.crit = -1;
            for (int g0 = 1; g0 < g; ++g0) {              // every way of sharing the range out; the members go, heaviest first, where the load per rank stays lowest
                if (g > 4 && g0 != g / 2 && g0 != g - g / 2 && g0 != 1 && g0 != g - 1) continue;
                Split sp; sp.load[0] = sp.load[1] = 0; sp.g[0] = g0; sp.g[1] = g - g0;
                for (int c : order) { const int b = (sp.load[0] + work[c]) / sp.g[0] <= (sp.load[1] + work[c]) / sp.g[1] ? 0 : 1; sp.bin[b].push_back(c); sp.load[b] += work[c]; }
                sp.crit = std::max(sp.load[0] / sp.g[0], sp.load[1] / sp.g[1]);
                if (best.crit < 0 || sp.crit < best.crit * (1.0 - 1e-12)) best = sp;
            }
            return best;
        };
        std::function<void(int, int)> own_subtree = [&](int root, int r) {
            vector<int> st(1, root);
            while (!st.empty()) { const int s = st.back(); st.pop_back(); S.sn_owner[s] = r; S.sn_glo[s] = r; S.sn_gsz[s] = 1; S.sn_gdepth[s] = 0;
                                  for (int q = S.child_ptr[s]; q < S.child_ptr[s + 1]; ++q) st.push_back(S.child_idx[q]); }
        };
        int maxdepth = 0;
        std::function<void(vector<int>, int, int, int)> assign = [&](vector<int> set, int a, int g, int depth) {
            if (set.empty()) return;
            if (g == 1) { for (int c : set) own_subtree(c, a); return; }
            for (int guard = 0; guard < nsn; ++guard) {
                int heavy = set[0];
                for (int c : set) if (work[c] > work[heavy] || (work[c] == work[heavy] && c < heavy)) heavy = c;
                const bool can_open = S.child_ptr[heavy + 1] > S.child_ptr[heavy];
                bool open = false;
                if (set.size() == 1) open = can_open;
                else if (can_open) {
                    vector<int> set2;
                    for (int c : set) if (c != heavy) set2.push_back(c);
                    for (int q = S.child_ptr[heavy]; q < S.child_ptr[heavy + 1]; ++q) set2.push_back(S.child_idx[q]);
                    open = own[heavy] + split_set(set2, g).crit < 0.98 * split_set(set, g).crit;
                }
                if (!open) break;
                S.sn_owner[heavy] = -1; S.sn_glo[heavy] = a; S.sn_gsz[heavy] = g; S.sn_gdepth[heavy] = depth; maxdepth = std::max(maxdepth, depth);
                set.erase(std::find(set.begin(), set.end(), heavy));
                for (int q = S.child_ptr[heavy]; q < S.child_ptr[heavy + 1]; ++q) set.push_back(S.child_idx[q]);
                if (set.empty()) return;
            }
            if (set.size() == 1) { own_subtree(set[0], a); return; }      // a leaf front nobody can split: one rank takes it
            Split sp = split_set(set, g);
            assign(sp.bin[0], a, sp.g[0], depth + 1);
            assign(sp.bin[1], a + sp.g[0], sp.g[1], depth + 1);
        };
        vector<int> roots;
        for (int s = 0; s < nsn; ++s) if (S.sn_parent[s] < 0) roots.push_back(s);
        assign(roots, 0, opt.nranks, 0);
        S.num_gdepths = maxdepth + 1;
    } else if (opt.nranks > 1) {
        // subtree work estimates
        vector<double> work(nsn, 0.0);
        for (int s = 0; s < nsn; ++s) {
            double k = S.sn_colptr[s + 1] - S.sn_colptr[s], m = S.sn_rowptr[s + 1] - S.sn_rowptr[s];
            work[s] += k * m * m + 64.0;   // flops-ish + launch overhead weight
            if (S.sn_parent[s] >= 0) work[S.sn_parent[s]] += work[s];
        }
        // grow a frontier of subtree roots from the tree roots until there are enough, balanced pieces
        std::set<std::pair<double,int>, std::greater<std::pair<double,int>>> front;
        double total = 0;
        for (int s = 0; s < nsn; ++s) if (S.sn_parent[s] < 0) { front.insert({work[s], s}); total += work[s]; }
        const double target = total / opt.nranks;
        auto total_sub = [](const std::set<std::pair<double,int>, std::greater<std::pair<double,int>>>& f) { double t = 0; for (auto& e : f) t += e.first; return t; };
        int guard = 0;
        while (!front.empty() && guard++ < nsn) {
            auto top = *front.begin();
            // split the heaviest frontier subtree until there are >= 2 pieces per rank and none is heavier than 60% of a
            // rank's share: deeper cuts only move work into the REPLICATED top, which every rank repeats (Amdahl)
            bool enough = ((int)front.size() >= 2 * opt.nranks && top.first <= 0.6 * target) ||
                          ((int)front.size() >= opt.nranks && top.first <= 1.15 * (total_sub(front) / opt.nranks));
            if (enough) break;
            int s = top.second;
            if (S.child_ptr[s + 1] == S.child_ptr[s]) {   // leaf: cannot split; stop if it is the biggest
                break;
            }
            front.erase(front.begin());
            // s moves to the replicated top; its children join the frontier
            for (int q = S.child_ptr[s]; q < S.child_ptr[s + 1]; ++q) front.insert({work[S.child_idx[q]], S.child_idx[q]});
        }
        // greedy LPT assignment of frontier subtrees to ranks
        vector<double> load(opt.nranks, 0.0);
        vector<int> root_owner(nsn, -2);
        for (auto& e : front) { int r = (int)(std::min_element(load.begin(), load.end()) - load.begin()); load[r] += e.first; root_owner[e.second] = r; }
        // propagate ownership down (parents have larger indices than children)
        for (int s = nsn - 1; s >= 0; --s) {
            if (root_owner[s] >= 0) S.sn_owner[s] = root_owner[s];
            else if (S.sn_parent[s] >= 0 && S.sn_owner[S.sn_parent[s]] >= 0) S.sn_owner[s] = S.sn_owner[S.sn_parent[s]];
            else S.sn_owner[s] = -1;
        }
        for (int s = 0; s < nsn; ++s) if (S.sn_owner[s] >= 0) { S.sn_glo[s] = S.sn_owner[s]; S.sn_gsz[s] = 1; }
    }
    lap("ownership");
    // ---- 12. storage: panels, contribution blocks, in-place separator chains (needs the ownership map) ----
    S.panel_off.assign(nsn, 0); S.cb_off.assign(nsn, 0); S.sn_ldp.assign(nsn, 0); S.sn_ldt.assign(nsn, 0); S.alias_child.assign(nsn, -1);
    {
        auto K = [&](int s) { return (int64_t)(S.sn_colptr[s + 1] - S.sn_colptr[s]); };
        auto Mf = [&](int s) { return (int64_t)(S.sn_rowptr[s + 1] - S.sn_rowptr[s]); };
        // (measured and not kept, r05: the TAIL links of the root chain -- synth_1e6: (146, 63) -> (83, 63) -> (20, 20), strict pivot loops of one workgroup
        // at the very end, 104 + 29 us; MBndryCntrl1 N = 100: 141 + 82 us of 1 350 -- classed BIG so that they join their predecessor's chain group.  A
        // BIG front's pivot block only sees its own k x k block (the rows below are tested a posteriori), a small front's strict loop the whole column:
        // on the dense hostile grid at u = 0.01 the delayed-pivot loop then needed a ninth edit for one column of the tail.  The pivoting of the last
        // fronts of the tree is not worth 0.5 % of a factorisation.)
        for (int s = 0; s < nsn; ++s) if (S.sn_class[s] == FC_BIG)
            for (int q = S.child_ptr[s]; q < S.child_ptr[s + 1]; ++q) {
                const int c = S.child_idx[q];
                if (S.sn_class[c] == FC_BIG && Mf(c) - K(c) == Mf(s) && S.sn_owner[c] == S.sn_owner[s] && S.sn_glo[c] == S.sn_glo[s] && S.sn_gsz[c] == S.sn_gsz[s]) { S.alias_child[s] = c; break; }
            }
        int64_t loff = 0, coff = 0;
        for (int s = 0; s < nsn; ++s) if (S.alias_child[s] < 0) loff += Mf(s) * K(s);
        S.l_doubles = loff;
        loff = 0;
        // ---- 12a. RECYCLING of the contribution blocks that only carry a contribution to their parent (round 6; what MUMPS' stack does, IpMumpsSolverInterface.cpp:151-177
        //      ICNTL(14) sizes exactly that workspace).  The addresses stay STATIC (the factorisation remains one replayable launch sequence): the plan is made here, once,
        //      over the LEVEL schedule.  A block of a BIG front that hosts no in-place chain is written at its front's level and dead once the front that extend-adds it (its
        //      parent in the assembly tree) has been formed -- plus a WINDOW of levels, because launches of a level may still run on the look-ahead streams while the next
        //      levels are enqueued: the numeric schedule joins those streams into the main one at every level that is a multiple of the window (numeric.hip, enqueue_factor),
        //      so whatever was launched at level l has completed before anything of level >= l + window starts.  Blocks that host a chain hold that chain's panels (factor
        //      storage) and stay; blocks of small fronts stay too (a few per cent of the pool; the data-flow launches over several levels tag them by epoch).  First fit over
        //      the free list, blocks sorted by (level, size descending).  One GPU only (a rank's own subtrees report to the arena after ALL its levels).
        //      MI355X_KKT_RECYCLE = 0 never / 1 always / unset: where it saves at least a quarter of the blocks and 1 GiB (synth_1e6: nothing to gain, 8 GiB, 4.7 of them hosts;
        //      MBndryCntrl_3D 78: 68 -> ~35 GiB). ----
        S.cb_window = 0; S.cb_resident_doubles = 0;
        vector<char> recyc(nsn, 0);
        int64_t plain_total = 0, cand_total = 0;
        {
            vector<char> host(nsn, 0);
            for (int s = 0; s < nsn; ++s) if (S.alias_child[s] >= 0) host[S.alias_child[s]] = 1;
            for (int s = 0; s < nsn; ++s) if (S.alias_child[s] < 0) {
                const int64_t mu = Mf(s) - K(s);
                plain_total += mu * mu;
                if (!host[s] && S.sn_class[s] == FC_BIG && mu > 0 && S.sn_parent[s] >= 0) { recyc[s] = 1; cand_total += mu * mu; }
            }
        }
        int mode = -1;      // auto
        if (const char* e = getenv("MI355X_KKT_RECYCLE")) mode = atoi(e) != 0 ? 1 : 0;
        if (opt.nranks > 1 || getenv("MI355X_KKT_FORCE_MULTI") || getenv("MI355X_KKT_POOL_PIECE_MIB")) mode = 0;      // (the multi-rank schedule has no periodic join of its streams; pieces are cut between blocks that do not overlap)
        const int WINDOW = 8;
        vector<int64_t> roff(nsn, -1);
        int64_t rpeak = 0;
        if (mode != 0 && cand_total > 0) {
            // the consuming level of a block: the level of the front whose child list holds it (the ASSEMBLY tree: a rehung side child is consumed where it was hung)
            vector<int> cons(nsn, S.num_levels);
            for (int p = 0; p < nsn; ++p) for (int q = S.child_ptr[p]; q < S.child_ptr[p + 1]; ++q) cons[S.child_idx[q]] = S.sn_level[p];
            vector<int> order;
            for (int s = 0; s < nsn; ++s) if (recyc[s]) order.push_back(s);
            std::sort(order.begin(), order.end(), [&](int a, int b) {
                if (S.sn_level[a] != S.sn_level[b]) return S.sn_level[a] < S.sn_level[b];
                const int64_t ma = Mf(a) - K(a), mb = Mf(b) - K(b);
                return ma != mb ? ma > mb : a < b; });
            // live blocks ordered by the level from which their space may be written again; free list ordered by offset (coalesced)
            std::multimap<int, std::pair<int64_t, int64_t>> live;      // free-from level -> (offset, size)
            std::map<int64_t, int64_t> freel;                          // offset -> size
            auto give_back = [&](int64_t off, int64_t sz) {
                auto it = freel.lower_bound(off);
                if (it != freel.begin()) { auto pr = std::prev(it); if (pr->first + pr->second == off) { off = pr->first; sz += pr->second; freel.erase(pr); } }
                if (it != freel.end() && off + sz == it->first) { sz += it->second; freel.erase(it); }
                freel[off] = sz;
            };
            for (int s : order) {
                const int lv = S.sn_level[s];
                while (!live.empty() && live.begin()->first <= lv) { give_back(live.begin()->second.first, live.begin()->second.second); live.erase(live.begin()); }
                const int64_t mu = Mf(s) - K(s), need = mu * mu;
                int64_t off = -1;
                for (auto it = freel.begin(); it != freel.end(); ++it)
                    if (it->second >= need) { off = it->first; const int64_t rest = it->second - need; freel.erase(it); if (rest > 0) freel[off + need] = rest; break; }
                if (off < 0) {      // grow: a free tail is extended rather than skipped
                    if (!freel.empty() && std::prev(freel.end())->first + std::prev(freel.end())->second == rpeak) { off = std::prev(freel.end())->first; freel.erase(std::prev(freel.end())); }
                    else off = rpeak;
                    rpeak = off + need;
                }
                roff[s] = off;
                // written at level lv, read at level cons[s]; everything launched there has completed before level cons[s] + WINDOW starts being enqueued -- rounded up to the
                // next join: free from the first multiple of WINDOW that is >= cons + 1 ... + WINDOW covers it (see enqueue_factor)
                live.insert({std::min(cons[s], S.num_levels) + WINDOW + 1, {off, need}});
            }
            const int64_t saved = cand_total - rpeak;
            if (mode == 1 || (4 * saved >= plain_total && 8 * saved >= (int64_t)1 << 30)) S.cb_window = WINDOW;
        }
        if (S.cb_window == 0) std::fill(recyc.begin(), recyc.end(), (char)0);
        for (int s = 0; s < nsn; ++s) {
            const int64_t k = K(s), m = Mf(s), mu = m - k;
            if (S.alias_child[s] >= 0) continue;
            S.panel_off[s] = loff; loff += m * k; S.sn_ldp[s] = (int)m; S.sn_ldt[s] = (int)mu;
            if (!recyc[s]) { S.cb_off[s] = coff; coff += mu * mu; }
        }
        S.cb_resident_doubles = coff;
        if (S.cb_window > 0) { for (int s = 0; s < nsn; ++s) if (recyc[s]) S.cb_off[s] = coff + roff[s]; coff += rpeak; }
        for (int s = 0; s < nsn; ++s) {
            const int ac = S.alias_child[s];
            if (ac < 0) continue;
            const int64_t k = K(s);        // children precede parents, so the child's placement is final
            const int64_t ldc = S.sn_ldt[ac];
            S.panel_off[s] = S.l_doubles + S.cb_off[ac]; S.sn_ldp[s] = (int)ldc;
            S.cb_off[s] = S.cb_off[ac] + k * ldc + k; S.sn_ldt[s] = (int)ldc;
        }
        S.cb_doubles = coff; S.cb_plain_doubles = plain_total;
        // chain groups: up to `chain_group` consecutive links of an in-place chain (<= 256 columns, consecutive levels) are
        // ONE unit for the trailing update (a single rank-(sum k) update at the last link; the links before it only update
        // the group's own remaining columns) and for the triangular solves (one launch pair per group).
        const int gmax = std::max(1, std::min(opt.chain_group, 4));
        S.solve_group = opt.solve_group;
        S.grp_pos.assign(nsn, 0); S.grp_rem.assign(nsn, 0);
        {   // the latency-bound top of the tree: the levels from which on no level has more than `maxch` BIG fronts
            int maxch = 8;
            maxch = (int)std::max(0ll, knob_int("grp_maxchains", maxch));
            vector<int> nbig(S.num_levels, 0);
            for (int s = 0; s < nsn; ++s) if (S.sn_class[s] == FC_BIG) nbig[S.sn_level[s]]++;
            S.grp_cut_level = S.num_levels;
            for (int lv = S.num_levels - 1; lv >= 0 && nbig[lv] <= maxch; --lv) S.grp_cut_level = lv;
        }
        vector<int> gcols(nsn, 0), alias_parent(nsn, -1);
        for (int s = 0; s < nsn; ++s) {
            const int ac = S.alias_child[s];
            gcols[s] = (int)K(s);
            // (a link joins its chain child's group only if that child is its ONLY child: nothing but the chain itself writes into the
            //  front, so a whole group can be factored in one launch sequence at the level of its first link -- numeric.hip, k_grp_*)
            if (ac >= 0 && S.grp_pos[ac] + 1 < gmax && gcols[ac] + K(s) <= 256 && S.sn_level[s] == S.sn_level[ac] + 1 &&
                S.child_ptr[s + 1] - S.child_ptr[s] == 1 && !(S.sn_level[s] >= S.grp_cut_level && S.sn_level[ac] < S.grp_cut_level)) {
                S.grp_pos[s] = S.grp_pos[ac] + 1; gcols[s] = gcols[ac] + (int)K(s); alias_parent[ac] = s;
            }
        }
        for (int s = nsn - 1; s >= 0; --s) { const int p = alias_parent[s]; if (p >= 0) S.grp_rem[s] = S.grp_rem[p] + (int)K(p); }
        // forward-solve vectors: an in-place chain shares ONE vector (the parent's entries are the child's update entries)
        S.cv_off.assign(nsn, 0);
        int64_t cvo = 0;
        for (int s = 0; s < nsn; ++s) {
            const int ac = S.alias_child[s];
            if (ac >= 0) S.cv_off[s] = S.cv_off[ac] + K(ac);
            else { S.cv_off[s] = cvo; cvo += Mf(s); }
        }
        S.cvec_doubles = cvo;
        // W = L*D panels of the BIG fronts: per-level scratch in banks (level mod 8) so that the panels of a chain group
        // (<= 4 consecutive levels) are all alive at the group's trailing update; partial sums of the backward dot products
        S.wb_off.assign(nsn, -1); S.gpart_off.assign(nsn, -1);
        vector<int64_t> lvl_used(S.num_levels, 0), lvl_part(S.num_levels, 0);
        for (int s = 0; s < nsn; ++s) if (S.sn_class[s] == FC_BIG) {
            const int lv = S.sn_level[s];
            S.wb_off[s] = lvl_used[lv]; lvl_used[lv] += Mf(s) * K(s);
            S.gpart_off[s] = lvl_part[lv]; lvl_part[lv] += ((Mf(s) - K(s) + 255) / 256 + 1) * (int64_t)gcols[s];
        }
        // 8 banks: the group after a look-ahead split (numeric.hip) must not overwrite the W panels its predecessor still reads
        int64_t bank[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bank_base[8];
        for (int lv = 0; lv < S.num_levels; ++lv) bank[lv & 7] = std::max(bank[lv & 7], lvl_used[lv]);
        S.wbuf_doubles = 0;
        for (int b = 0; b < 8; ++b) { bank_base[b] = S.wbuf_doubles; S.wbuf_doubles += bank[b]; }
        for (int s = 0; s < nsn; ++s) if (S.wb_off[s] >= 0) S.wb_off[s] += bank_base[S.sn_level[s] & 7];
        S.gpart_doubles = 0;
        for (int64_t u : lvl_part) S.gpart_doubles = std::max(S.gpart_doubles, u);
    }
    return true;
}

// =======================================================================================
bool analyse(Symbolic& S, const SymbolicOptions& opt, int n, int nnz, const int* ri, const int* ci,
             int format, const double* vals)
{
    BlockCache::Scope recycle;          // (declared first: destroyed last, after every array of the analysis has been returned)
    PoolScope workers(analysis_threads());
    double t0 = now_s(), tl = t0;
    auto lap = [&](const char* what) { if (opt.verbose >= 2) { double t = now_s(); fprintf(stderr, "[mi355x_kkt]   %-28s %.3f s\n", what, t - tl); tl = t; } };
    S = Symbolic();
    S.n = n; S.nnz_in = nnz;
    if (n < 0 || nnz < 0) { S.error = "analyse: negative size"; return false; }
    if (n == 0) { S.sn_colptr.assign(1, 0); S.sn_rowptr.assign(1, 0); S.acolptr.assign(1, 0); S.level_ptr.assign(1, 0); S.child_ptr.assign(1, 0); S.dup_ptr.assign(1, 0); return true; }

    // ---- 1. pattern ----
    Pattern P;
    if (!build_pattern(n, nnz, ri, ci, opt.index_base, format, P, S.error)) return false;
    vector<int> xadj, adj;
    build_adjacency(n, P.colptr, P.row, xadj, adj);

    lap("pattern+adjacency");
    // ---- 2. pairing ----
    vector<char> zrow(n, 0);
    if (opt.matching) zero_diag_matching(n, P, xadj, adj, vals, nnz, S.pair_of, S.num_pairs, zrow);
    else S.pair_of.assign(n, -1);

    lap("pairing");
    // ---- 3. compressed graph ----
    vector<int> cid(n, -1); int nc = 0;
    vector<int> cfirst; cfirst.reserve(n);   // representative (first member) of each compressed node
    for (int i = 0; i < n; ++i) if (cid[i] < 0) {
        cid[i] = nc; int q = S.pair_of[i];
        if (q >= 0) cid[q] = nc;
        cfirst.push_back(i); ++nc;
    }
    Graph CG; CG.n = nc; CG.xadj.assign(nc + 1, 0);
    {
        // neighbours of a compressed node = the compressed nodes of its members' neighbours, each once, in the order of first occurrence (the order the serial
        // construction with one mark array produced: the dissection walks these lists, so the order is part of the ordering).  Chunks of nodes on threads,
        // each with its own output, stitched together behind a prefix sum.
        const int T = analysis_threads();
        const int TC = (T <= 1 || nc < 4096) ? 1 : T;
        std::vector<vector<int>> part(TC);
        parallel_chunks(nc, TC, [&](long long cb, long long ce, int t) {
            vector<int>& out = part[t];
            out.reserve((size_t)((xadj[n] / std::max(nc, 1)) * (ce - cb) + 16));
            vector<int> cand, uniq; vector<char> used;
            for (int c = (int)cb; c < (int)ce; ++c) {
                const int mem[2] = { cfirst[c], S.pair_of[cfirst[c]] };
                const size_t o0 = out.size();
                int deg = 0;
                for (int k = 0; k < 2; ++k) if (mem[k] >= 0) deg += xadj[mem[k] + 1] - xadj[mem[k]];
                if (deg <= 48) {
                    for (int k = 0; k < 2; ++k) { const int i = mem[k]; if (i < 0) continue;
                        for (int p = xadj[i]; p < xadj[i + 1]; ++p) {
                            const int d = cid[adj[p]];
                            if (d == c) continue;
                            bool seen = false;
                            for (size_t q = o0; q < out.size(); ++q) if (out[q] == d) { seen = true; break; }
                            if (!seen) out.push_back(d);
                        } }
                } else {
                    cand.clear();
                    for (int k = 0; k < 2; ++k) { const int i = mem[k]; if (i < 0) continue; for (int p = xadj[i]; p < xadj[i + 1]; ++p) { const int d = cid[adj[p]]; if (d != c) cand.push_back(d); } }
                    uniq.assign(cand.begin(), cand.end());
                    std::sort(uniq.begin(), uniq.end()); uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
                    used.assign(uniq.size(), 0);
                    for (int d : cand) { const size_t u = std::lower_bound(uniq.begin(), uniq.end(), d) - uniq.begin(); if (!used[u]) { used[u] = 1; out.push_back(d); } }
                }
                CG.xadj[c + 1] = (int)(out.size() - o0);
            }
        });
        for (int c = 0; c < nc; ++c) CG.xadj[c + 1] += CG.xadj[c];
        CG.adj.resize((size_t)CG.xadj[nc]);
        parallel_chunks(nc, TC, [&](long long cb, long long, int t) {      // (the same chunks: thread t copies its own output)
            if (!part[t].empty()) std::memcpy(CG.adj.data() + CG.xadj[cb], part[t].data(), part[t].size() * sizeof(int));
        });
    }

    lap("compressed graph");
    // ---- 4. ordering ----
    vector<int> corder(nc);
    if (opt.ordering == 2) std::iota(corder.begin(), corder.end(), 0);
    else if (opt.ordering == 1) { NestedDissection nd(CG, opt.nd_leaf); vector<int> all(nc); std::iota(all.begin(), all.end(), 0); nd.mindeg().order(all, corder.data()); }
    else {
        // nested dissection; the recursion is task parallel: the main thread splits the largest pending piece until there
        // are enough pieces, then worker threads (each with private scratch) finish them -- output ranges are disjoint
        const int T = analysis_threads();
        NestedDissection nd(CG, opt.nd_leaf);
        vector<NestedDissection::Task> st;
        { vector<int> all(nc); std::iota(all.begin(), all.end(), 0); st.push_back({std::move(all), 0}); }
        std::fill(corder.begin(), corder.end(), -1);
        if (T <= 1 || nc < 50000) nd.drain(st, corder);
        else {
            // a shared pool of pieces: a worker takes the largest pending piece, bisects it (step) and puts the parts back, or -- once a piece is
            // small -- finishes it on its own (drain).  The top bisections therefore overlap as soon as there are two pieces; up to round 2 the
            // main thread made the first 8 T pieces alone, i.e. walked the whole graph ~8 times serially (0.5 of the 0.7 s of this phase at
            // n = 10^6).  What a piece becomes depends on the piece only, not on who handles it: the ordering does not depend on T.
            const int small_piece = std::max(nd.leaf(), std::max(20000, nc / (32 * T)));
            std::mutex mu; std::condition_variable cv;
            size_t outstanding = st.size();
            std::vector<std::thread> th;
            for (int w = 0; w < T; ++w) th.emplace_back([&] {
                NestedDissection local(CG, opt.nd_leaf);
                vector<NestedDissection::Task> mine;
                for (;;) {
                    NestedDissection::Task t;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return !st.empty() || outstanding == 0; });
                        if (st.empty()) return;
                        size_t big = 0; for (size_t q = 1; q < st.size(); ++q) if (st[q].nodes.size() > st[big].nodes.size()) big = q;
                        t = std::move(st[big]); st.erase(st.begin() + big);
                    }
                    mine.clear();
                    if ((int)t.nodes.size() <= small_piece) { mine.push_back(std::move(t)); local.drain(mine, corder); }
                    else local.step(std::move(t), mine, corder);
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        outstanding += mine.size(); --outstanding;
                        for (auto& c : mine) st.push_back(std::move(c));
                    }
                    cv.notify_all();
                }
            });
            for (auto& x : th) x.join();
        }
    }
    // expand: within a pair the non-zero-diagonal member (the "variable") comes first, the
    // zero-diagonal row (the "constraint") second.
    vector<int> perm; perm.reserve(n);
    for (int k = 0; k < nc; ++k) {
        int a = cfirst[corder[k]], b = S.pair_of[a];
        if (b < 0) perm.push_back(a);
        else { if (zrow[a]) std::swap(a, b); perm.push_back(a); perm.push_back(b); }
    }
    vector<int> iperm(n);
    for (int k = 0; k < n; ++k) iperm[perm[k]] = k;

    lap("ordering (ND+MD)");
    // ---- 5. elimination tree + postorder (pairs stay adjacent) ----
    vector<int> parent(n, -1);
    {
        vector<int> anc(n, -1);
        for (int k = 0; k < n; ++k) {
            int i0 = perm[k];
            for (int p = xadj[i0]; p < xadj[i0 + 1]; ++p) {
                int r = iperm[adj[p]];
                if (r >= k) continue;
                while (anc[r] != -1 && anc[r] != k) { int nx = anc[r]; anc[r] = k; r = nx; }
                if (anc[r] == -1) { anc[r] = k; parent[r] = k; }
            }
        }
    }
    vector<int> post(n);   // post[newnew] = new
    {
        vector<int> head(n, -1), next(n, -1);
        // children in DEscending insertion so that lists come out ascending; the paired child is moved last
        for (int k = n - 1; k >= 0; --k) if (parent[k] >= 0) { next[k] = head[parent[k]]; head[parent[k]] = k; }
        // move paired child (k-1 of k when they form a pair) to the end of k's child list
        for (int k = 1; k < n; ++k) {
            if (S.pair_of[perm[k]] == perm[k - 1] && parent[k - 1] == k) {
                // remove k-1 from list of k and append at the tail
                int prev = -1, c = head[k];
                while (c != -1 && c != k - 1) { prev = c; c = next[c]; }
                if (c == k - 1 && next[c] != -1) {
                    if (prev == -1) head[k] = next[c]; else next[prev] = next[c];
                    int tail = head[k]; while (next[tail] != -1) tail = next[tail];
                    next[tail] = c; next[c] = -1;
                }
            }
        }
        int cnt = 0; vector<int> stack; stack.reserve(64);
        vector<int> roots; for (int k = 0; k < n; ++k) if (parent[k] < 0) roots.push_back(k);
        for (int r : roots) {
            stack.push_back(r);
            while (!stack.empty()) {
                int v = stack.back(); int c = head[v];
                if (c == -1) { post[cnt++] = v; stack.pop_back(); }
                else { head[v] = next[c]; stack.push_back(c); }
            }
        }
    }
    {   // compose permutations, relabel parent
        vector<int> perm2(n), newlab(n);
        for (int t = 0; t < n; ++t) { perm2[t] = perm[post[t]]; newlab[post[t]] = t; }
        vector<int> parent2(n, -1);
        for (int k = 0; k < n; ++k) if (parent[k] >= 0) parent2[newlab[k]] = newlab[parent[k]];
        perm.swap(perm2); parent.swap(parent2);
        for (int k = 0; k < n; ++k) iperm[perm[k]] = k;
    }
    S.perm = perm; S.iperm = iperm;

    lap("etree+postorder");
    // ---- 6. permuted lower CSC + maps (re-run when the tree amalgamation of step 8b renumbers columns) ----
    auto build_permuted_csc = [&]()
    {
        const int nnzA = (int)P.row.size();
        const int T = analysis_threads();
        vector<int> pc(nnzA), pr(nnzA);
        parallel_chunks(n, T, [&](long long jb, long long je, int) {
            for (int j = (int)jb; j < (int)je; ++j)
                for (int p = P.colptr[j]; p < P.colptr[j + 1]; ++p) {
                    const int a = iperm[P.row[p]], b = iperm[j];
                    pc[p] = std::min(a, b); pr[p] = std::max(a, b);
                }
        });
        // place every entry into its (new) column, then sort the rows of each column -- columns are independent
        vector<int> cnt, src;
        parallel_bucket(nnzA, n, T, [&](long long p) { return pc[p]; }, cnt, src);
        S.acolptr = cnt;
        vector<int> old2new(nnzA), new2old(nnzA); S.arow.resize(nnzA); S.acol.resize(nnzA);
        parallel_chunks(n, T, [&](long long jb, long long je, int) {
            vector<std::pair<int,int>> tmp;
            for (int j = (int)jb; j < (int)je; ++j) {
                const int q0 = cnt[j], q1 = cnt[j + 1];
                tmp.clear();
                for (int q = q0; q < q1; ++q) tmp.emplace_back(pr[src[q]], src[q]);
                if (q1 - q0 > 1) std::sort(tmp.begin(), tmp.end());
                for (int q = q0; q < q1; ++q) { S.arow[q] = tmp[q - q0].first; S.acol[q] = j; old2new[tmp[q - q0].second] = q; new2old[q] = tmp[q - q0].second; }
            }
        });
        S.nnz_a = nnzA;
        S.trip2slot.resize(nnz);
        parallel_chunks(nnz, T, [&](long long tb, long long te, int) { for (long long t = tb; t < te; ++t) S.trip2slot[t] = old2new[P.t2slot[t]]; });
        // duplicate lists: the triplets of every slot in ascending order (the device sums them in this order => bitwise reproducible).  The
        // pattern pass left them as runs of its (column, row, triplet)-sorted list: a prefix sum over the slots in their new order and a copy
        // (no second bucketing of the triplets, no sort)
        S.dup_ptr.assign((size_t)nnzA + 1, 0);
        if ((int)P.scnt.size() == nnzA && (int)P.tsorted.size() == nnz) {
            for (int q = 0; q < nnzA; ++q) S.dup_ptr[q + 1] = S.dup_ptr[q] + P.scnt[new2old[q]];
            S.dup_src.resize((size_t)nnz);
            parallel_chunks(nnzA, T, [&](long long qb, long long qe, int) {
                for (long long q = qb; q < qe; ++q) { const int o = new2old[q]; const int* src = P.tsorted.data() + P.sfirst[o]; int* dst = S.dup_src.data() + S.dup_ptr[q];
                                                      for (int e = 0; e < P.scnt[o]; ++e) dst[e] = src[e]; }
            });
        } else {
            parallel_bucket(nnz, nnzA, T, [&](long long t) { return S.trip2slot[t]; }, S.dup_ptr, S.dup_src);
            parallel_chunks(nnzA, T, [&](long long qb, long long qe, int) {
                for (long long q = qb; q < qe; ++q) if (S.dup_ptr[q + 1] - S.dup_ptr[q] > 1) std::sort(S.dup_src.begin() + S.dup_ptr[q], S.dup_src.begin() + S.dup_ptr[q + 1]);
            });
        }
    };
    build_permuted_csc();
    lap("permuted CSC + maps");
    // ---- 7. column counts (skeleton matrix / least common ancestors; matrix is postordered) ----
    vector<int> cc(n, 0);
    {
        vector<int> first(n, -1), maxfirst(n, -1), prevleaf(n, -1), anc(n), delta(n, 0);
        for (int k = 0; k < n; ++k) { anc[k] = k; int j = k; delta[j] = (first[j] == -1) ? 1 : 0; for (; j != -1 && first[j] == -1; j = parent[j]) first[j] = k; }
        for (int j = 0; j < n; ++j) {
            if (parent[j] != -1) delta[parent[j]]--;
            for (int p = S.acolptr[j]; p < S.acolptr[j + 1]; ++p) {
                int i = S.arow[p];
                if (i <= j || first[j] <= maxfirst[i]) continue;
                maxfirst[i] = first[j];
                int jprev = prevleaf[i]; prevleaf[i] = j;
                if (jprev == -1) { delta[j]++; continue; }
                int q = jprev; while (q != anc[q]) q = anc[q];
                for (int s = jprev; s != q;) { int sp = anc[s]; anc[s] = q; s = sp; }
                delta[j]++; delta[q]--;
            }
            if (parent[j] != -1) anc[j] = parent[j];
        }
        for (int j = 0; j < n; ++j) cc[j] = delta[j];
        for (int j = 0; j < n; ++j) if (parent[j] != -1) cc[parent[j]] += cc[j];
    }

    lap("column counts");
    // ---- 8. supernodes: fundamental + forced pairs, then relaxed amalgamation of chains ----
    const int maxcols = std::max(2, opt.max_sn_cols);
    // fronts of order >= wide_from (the separator chains of the blocked path) get panels of 2*maxcols columns: half as many
    // tree levels and twice the arithmetic intensity of the Schur update (the pivot-block kernels handle <= 128 columns)
    const int wide_from = opt.wide_panels > 1 ? opt.wide_panels : (opt.wide_panels == 1 ? 512 : (1 << 30));     // > 1: the front order from which panels are wide     // default off: measured slower (pivot block of 128 costs more than it saves)
    // (a) whole small subtrees become one dense supernode: in the latency-bound regime (fronts of a few rows) dense
    //     arithmetic on <= leaf_cols columns is free, while every tree level costs a kernel launch and a dependent
    //     HBM round trip.  A subtree is a contiguous column range in the postorder.
    vector<char> forced_join(n, 0), forced_start(n, 0);
    if (opt.leaf_cols > 1) {
        const int cap_m = 64;     // the merged front must still fit the one-wavefront LDS kernel
        vector<int> desc(n, 1);
        for (int j = 0; j < n; ++j) if (parent[j] >= 0) desc[parent[j]] += desc[j];
        auto mergeable = [&](int j) { return desc[j] <= opt.leaf_cols && desc[j] + cc[j] - 1 <= cap_m; };
        for (int j = 0; j < n; ++j) {
            if (desc[j] < 2 || !mergeable(j)) continue;
            if (parent[j] >= 0 && mergeable(parent[j])) continue;      // not maximal
            const int first = j - desc[j] + 1;
            forced_start[first] = 1;
            for (int c = first + 1; c <= j; ++c) forced_join[c] = 1;
        }
    }
    vector<int> fstart;   // first column of each fundamental supernode
    {
        int len = 0;
        for (int j = 0; j < n; ++j) {
            bool join = false;
            if (forced_join[j]) join = true;
            else if (!forced_start[j] && j > 0 && parent[j - 1] == j) {
                bool pair = (S.pair_of[perm[j]] == perm[j - 1]);
                if (pair) join = true;
                else if (cc[j - 1] == cc[j] + 1 && len < (cc[j] >= wide_from ? 2 * maxcols : maxcols) - 1) join = true;   // -1: a forced pair may still add one column
            }
            if (!join) { fstart.push_back(j); len = 1; } else ++len;
        }
    }
    {
        // relaxed amalgamation with a stack of merged supernodes (k cols, m front order, z explicit zeros)
        struct M { int c0, k, m; double z; };
        vector<M> st;
        int nf = (int)fstart.size();
        auto zfrac = [&](int k) { return k <= 16 ? 0.5 : (k <= 48 ? 0.15 : 0.05); };
        for (int f = 0; f < nf; ++f) {
            int c0 = fstart[f], c1 = (f + 1 < nf) ? fstart[f + 1] : n;
            M cur{c0, c1 - c0, (c1 - c0) + cc[c1 - 1] - 1, 0.0};   // front order = columns + update rows of the last column
            // try to absorb preceding merged supernodes that are children of cur
            while (!st.empty()) {
                M& ch = st.back();
                int lastc = ch.c0 + ch.k - 1;
                int pcol = parent[lastc];
                if (pcol < cur.c0 || pcol >= cur.c0 + cur.k) break;          // not a child of cur
                int knew = ch.k + cur.k;
                if (knew > (cur.m >= wide_from ? 2 * maxcols : maxcols)) break;
                int mnew = ch.k + cur.m;
                double znew = ch.z + cur.z + (double)ch.k * (double)(ch.k + cur.m - ch.m);
                double total = (double)knew * mnew - 0.5 * knew * (knew - 1);
                bool ok = (knew <= opt.nemin) || (znew <= zfrac(knew) * total);
                if (!ok) break;
                cur = M{ch.c0, knew, mnew, znew};
                st.pop_back();
            }
            st.push_back(cur);
        }
        S.num_sn = (int)st.size();
        S.sn_colptr.resize(S.num_sn + 1);
        for (int s = 0; s < S.num_sn; ++s) S.sn_colptr[s] = st[s].c0;
        S.sn_colptr[S.num_sn] = n;
    }
    // ---- 8b. tree amalgamation: merge small sibling/child supernodes into their parent even when their columns are
    //      not contiguous (the chain pass above only joins neighbours).  In the latency-bound regime every tree level is
    //      a kernel launch plus a few dependent HBM round trips, so a 3-separator node of ~16 columns beats three levels of
    //      ~5 columns.  Columns are renumbered by a postorder of the merged tree (any topological order of the elimination
    //      tree is an equivalent elimination order), then the permuted pattern is rebuilt once.
    {
        const bool on = opt.tree_merge > 0 || (opt.tree_merge < 0 && n <= 400000);
        const int nsn0 = S.num_sn;
        if (on && nsn0 > 1) {
            const int KM = 20, MM = 32;
            vector<int> snof0(n), par0(nsn0, -1), gk(nsn0), gm(nsn0), grp(nsn0);
            for (int s = 0; s < nsn0; ++s) for (int j = S.sn_colptr[s]; j < S.sn_colptr[s + 1]; ++j) snof0[j] = s;
            for (int s = 0; s < nsn0; ++s) {
                const int last = S.sn_colptr[s + 1] - 1;
                gk[s] = S.sn_colptr[s + 1] - S.sn_colptr[s]; gm[s] = gk[s] + cc[last] - 1; grp[s] = s;
                if (parent[last] >= 0) par0[s] = snof0[parent[last]];
            }
            int merges = 0;
            for (int s = 0; s < nsn0; ++s) {
                const int p = par0[s];
                if (p < 0) continue;
                const int newk = gk[s] + gk[p], newm = gk[s] + gm[p];
                if (newk > KM || newm > MM) continue;
                const double z = (double)gk[s] * (double)(newm - gm[s]);
                if (z > 0.6 * (double)newk * newm) continue;
                grp[s] = p; gk[p] = newk; gm[p] = newm; ++merges;
            }
            if (merges > 0) {
                auto find = [&](int s) { while (grp[s] != s) s = grp[s]; return s; };
                vector<int> rep(nsn0);
                for (int s = nsn0 - 1; s >= 0; --s) rep[s] = (grp[s] == s) ? s : rep[grp[s]];     // parents have larger indices
                (void)find;
                // members of each group (ascending) and child groups
                vector<int> mhead(nsn0, -1), mnext(nsn0, -1), chead(nsn0, -1), cnext(nsn0, -1);
                for (int s = nsn0 - 1; s >= 0; --s) { const int g = rep[s]; mnext[s] = mhead[g]; mhead[g] = s; }
                vector<int> roots;
                for (int s = nsn0 - 1; s >= 0; --s) if (rep[s] == s) {
                    // parent group = group of the parent of the group's top member s
                    const int p = par0[s];
                    if (p < 0) roots.push_back(s); else { const int pg = rep[p]; cnext[s] = chead[pg]; chead[pg] = s; }
                }
                // postorder over groups (child lists are ascending because of the descending insertion), emit columns
                vector<int> newpos; newpos.reserve(n);            // newpos[t] = old column at new position t
                vector<int> newstart;                             // first new column of each group
                vector<int> stack, itc(nsn0);
                for (int s = 0; s < nsn0; ++s) itc[s] = chead[s];
                std::sort(roots.begin(), roots.end());
                for (int r : roots) {
                    stack.push_back(r);
                    while (!stack.empty()) {
                        const int g = stack.back();
                        if (itc[g] != -1) { const int c = itc[g]; itc[g] = cnext[c]; stack.push_back(c); continue; }
                        stack.pop_back();
                        newstart.push_back((int)newpos.size());
                        for (int ms = mhead[g]; ms != -1; ms = mnext[ms])
                            for (int j = S.sn_colptr[ms]; j < S.sn_colptr[ms + 1]; ++j) newpos.push_back(j);
                    }
                }
                // renumber: perm, iperm; the column etree / counts are not needed any more
                vector<int> perm2(n);
                for (int t = 0; t < n; ++t) perm2[t] = perm[newpos[t]];
                perm.swap(perm2);
                for (int t = 0; t < n; ++t) iperm[perm[t]] = t;
                S.perm = perm; S.iperm = iperm;
                S.num_sn = (int)newstart.size();
                S.sn_colptr = newstart; S.sn_colptr.push_back(n);
                build_permuted_csc();
            }
        }
    }
    if (!finish_analysis(S, opt, lap)) return false;
    S.time_analyse = now_s() - t0;
    if (opt.verbose)
        fprintf(stderr, "[mi355x_kkt] analyse: n=%d nnzA=%d pairs=%d nsn=%d levels=%d maxfront=%d maxsn=%d nnzL=%lld flops=%.3g big=%d rehung=%d  %.3fs\n",
                n, S.nnz_a, S.num_pairs, S.num_sn, S.num_levels, S.maxfront, S.maxsupernode, (long long)S.nnz_l, (double)S.flops_factor, S.num_big, S.num_rehung, S.time_analyse);
    return true;
}

// =======================================================================================
// Delayed pivoting ACROSS fronts, as an edit of the supernode partition.
//
// What MA27 / MA57 / MA97 / MUMPS / SPRAL do when a fully-summed column of a front finds no pivot that passes the threshold
// test: the column stays uneliminated, becomes part of the contribution block and is a fully-summed column of the PARENT front,
// where more of its row is summed and more partners are available (Duff & Reid 1983; the adapters read the consequences:
// IpMa97SolverInterface.cpp:719-779 info.num_delay, IpMa27TSolverInterface.cpp:565-622 workspace growth + refactorisation).
// Here the device structures are static (addresses, launch lists, hipGraphs), so a delay is a change of the STRUCTURE: the
// failed columns of a front leave its supernode and join the parent's, the symbolic structures are rebuilt from the edited
// partition (no new ordering: the permutation only moves the delayed columns behind their former siblings), the numeric side
// is set up again and the matrix is refactored -- MA27's "grow the workspace and call MA27BD again" in our terms.  The edited
// structure is kept for the following factorisations (same sparsity, similar numbers: Ipopt's next iteration).
//
// `marked`: columns (CURRENT permuted numbering) the last factorisation could not pivot (bit 1 of DevView::zpiv).  ALL of them move, whatever
// lies beneath them: measured on the hostile sets (tools/hostile_delay_run.py), waiting for the fronts below a mark to come clean first
// ("a forced pivot poisons what its ancestors see") makes the marks of a separator chain climb ONE link per refactorisation -- 22-30 rounds
// where 4-8 do with everything moving at once, at +10 % nnz(L); a column delayed without need costs fill, not correctness.  Root fronts have
// nowhere to delay to.  A parent that would exceed max_sn_cols columns is cut into a chain of supernodes (delayed columns first, together
// with as many of the parent's own as fit).
// =======================================================================================
bool restructure_delays(const Symbolic& C, const SymbolicOptions& opt, const std::vector<int>& marked, const std::vector<int>& hops, Symbolic& S, int* moved_out, std::vector<char>* acted)
{
    BlockCache::Scope recycle;
    PoolScope workers(analysis_threads());
    const double t0 = now_s(); double tl = t0;
    auto lap = [&](const char* what) { if (opt.verbose >= 2) { double t = now_s(); fprintf(stderr, "[mi355x_kkt]   (delay) %-20s %.3f s\n", what, t - tl); tl = t; } };
    const int n = C.n, nsn0 = C.num_sn;
    if (moved_out) *moved_out = 0;
    S = Symbolic();
    if (n == 0 || nsn0 == 0) { S.error = "restructure: empty structure"; return false; }
    // ---- which marks count, and where their columns go ----
    vector<char> mk(n, 0);
    vector<int> nmark(nsn0, 0), target(n, -1);            // target[j]: the front column j moves to
    for (size_t q = 0; q < marked.size(); ++q) {
        const int j = marked[q];
        if (j >= 0 && j < n && !mk[j]) { mk[j] = 1; nmark[C.sn_of[j]]++; target[j] = q < hops.size() ? std::max(1, hops[q]) : 1; }      // (levels to climb, for now)
    }
    vector<int> incount(nsn0 + 1, 0);
    int moved = 0;
    // the parent in the ELIMINATION tree: the front that holds the first update row (finish_analysis 9b may have hung a side child of a chain link on a
    // lower link of the chain for the assembly; a delayed column must go where it is eliminated later than its siblings)
    auto nat_parent = [&](int s) { const int k = C.sn_colptr[s + 1] - C.sn_colptr[s]; return C.sn_rowptr[s] + k < C.sn_rowptr[s + 1] ? C.sn_of[C.sn_rows[C.sn_rowptr[s] + k]] : -1; };
    for (int s = 0; s < nsn0; ++s) {
        if (nmark[s] == 0) continue;
        const int p = nat_parent(s);
        for (int j = C.sn_colptr[s]; j < C.sn_colptr[s + 1]; ++j) if (mk[j]) {
            if (p < 0) { mk[j] = 0; continue; }                                             // a root front has nowhere to delay to
            int t = p;                                                                      // a column that failed before climbs several levels at once
            for (int h = target[j]; h > 1 && nat_parent(t) >= 0; --h) t = nat_parent(t);
            target[j] = t; ++moved;
        }
    }
    // A target front that is FULL (max_sn_cols columns) would be cut in two by an arrival: a 64-column link of a separator chain that receives one column became a
    // link of 64 and a link of ONE column -- a tree level of its own, pivot block + panel solve + rank-1 update over thousands of rows (round 5; measured in round 6: one
    // edit of 100 random columns made the factorisation of a 2 * 10^5 grid 36 % slower, 4.36 -> 5.94 ms).  Inside a chain -- the parent is the next supernode, its
    // front is this front's update rows -- the column may just as well wait one link further up, and further, until a link has room (the last link of a separator
    // is rarely full): a longer delay, no front displaced, no fill (the chain is dense).  Nobody else changes front, which the delayed-pivot loop of the hostile test
    // systems depends on (moving the boundary between two full links instead -- the link's own last column riding into the parent -- left forced pivots after 8 rounds).
    {
        const int maxcols = std::max(2, opt.max_sn_cols);
        vector<int> load(nsn0);
        for (int s = 0; s < nsn0; ++s) load[s] = C.sn_colptr[s + 1] - C.sn_colptr[s] - nmark[s];
        // (... where the full front is LARGE -- 256 rows and more: there the one-column link is a level of ~100 us launches, and it pushes every ancestor one
        //  level up, out of step with its siblings -- and the edit moves FEW columns, at most n / 64: an Ipopt run delays a handful per factorisation.  The hostile
        //  systems of the tests -- order 800, hundreds of columns per round, every column moving four times -- needed a ninth and tenth round with the extra hops.)
        for (int j = 0; j < n; ++j) if (mk[j]) {
            int t = target[j];
            for (int hop = 0; hop < 256 && moved <= n / 64 && load[t] >= maxcols && C.sn_rowptr[t + 1] - C.sn_rowptr[t] >= 256 && t + 1 < nsn0 && nat_parent(t) == t + 1; ++hop) ++t;
            target[j] = t; load[t]++; incount[t + 1]++;
        }
    }
    if (moved_out) *moved_out = moved;
    if (acted) { acted->assign(marked.size(), 0); for (size_t q = 0; q < marked.size(); ++q) { const int j = marked[q]; if (j >= 0 && j < n && mk[j]) (*acted)[q] = 1; } }
    if (moved == 0) { S.error = "restructure: nothing to delay"; return false; }
    for (int s = 0; s < nsn0; ++s) incount[s + 1] += incount[s];
    vector<int> incoming(moved), fillp(incount.begin(), incount.end() - 1);
    for (int j = 0; j < n; ++j) if (mk[j]) incoming[fillp[target[j]]++] = j;               // (ascending column order inside every target)
    // ---- the new partition: supernodes in their old order, [columns delayed into it] + [its own remaining columns], cut at maxcols ----
    const int maxcols = std::max(2, opt.max_sn_cols);
    vector<int> newpos; newpos.reserve(n);            // newpos[t] = column (current numbering) at new position t
    vector<int> ncolptr; ncolptr.reserve((size_t)nsn0 + 16);
    for (int s = 0; s < nsn0; ++s) {
        const int start = (int)newpos.size();
        for (int q = incount[s]; q < incount[s + 1]; ++q) newpos.push_back(incoming[q]);
        for (int j = C.sn_colptr[s]; j < C.sn_colptr[s + 1]; ++j) if (!mk[j]) newpos.push_back(j);
        const int cnt = (int)newpos.size() - start;
        for (int o = 0; o < cnt; o += maxcols) ncolptr.push_back(start + o);
    }
    if ((int)newpos.size() != n) { S.error = "restructure: internal error (partition does not cover the columns)"; return false; }
    const int nsn = (int)ncolptr.size();
    ncolptr.push_back(n);
    // ---- permutation ----
    S.n = n; S.nnz_in = C.nnz_in; S.nnz_a = C.nnz_a; S.num_pairs = C.num_pairs; S.pair_of = C.pair_of;
    vector<int> newlab(n);
    S.perm.resize(n); S.iperm.resize(n);
    for (int t = 0; t < n; ++t) { newlab[newpos[t]] = t; S.perm[t] = C.perm[newpos[t]]; }
    for (int t = 0; t < n; ++t) S.iperm[S.perm[t]] = t;
    lap("partition");
    // ---- permuted lower CSC from the current one: relabel, re-bucket, sort the rows of every column; slots keep their triplets ----
    {
        const int nnzA = C.nnz_a, T = analysis_threads();
        vector<int> pc(nnzA), pr(nnzA);
        parallel_chunks(nnzA, T, [&](long long qb, long long qe, int) {
            for (long long q = qb; q < qe; ++q) { const int a = newlab[C.arow[q]], b = newlab[C.acol[q]]; pc[q] = std::min(a, b); pr[q] = std::max(a, b); }
        });
        vector<int> cnt, src;
        parallel_bucket(nnzA, n, T, [&](long long q) { return pc[q]; }, cnt, src);
        S.acolptr = cnt;
        vector<int> old2new(nnzA), new2old(nnzA); S.arow.resize(nnzA); S.acol.resize(nnzA);
        parallel_chunks(n, T, [&](long long jb, long long je, int) {
            vector<std::pair<int,int>> tmp;
            for (int j = (int)jb; j < (int)je; ++j) {
                const int q0 = cnt[j], q1 = cnt[j + 1];
                tmp.clear();
                for (int q = q0; q < q1; ++q) tmp.emplace_back(pr[src[q]], src[q]);
                if (q1 - q0 > 1) std::sort(tmp.begin(), tmp.end());
                for (int q = q0; q < q1; ++q) { S.arow[q] = tmp[q - q0].first; S.acol[q] = j; old2new[tmp[q - q0].second] = q; new2old[q] = tmp[q - q0].second; }
            }
        });
        S.trip2slot.resize(C.trip2slot.size());
        parallel_chunks((long long)C.trip2slot.size(), T, [&](long long tb, long long te, int) { for (long long t = tb; t < te; ++t) S.trip2slot[t] = old2new[C.trip2slot[t]]; });
        S.dup_ptr.assign((size_t)nnzA + 1, 0);
        for (int q = 0; q < nnzA; ++q) S.dup_ptr[q + 1] = S.dup_ptr[q] + (C.dup_ptr[new2old[q] + 1] - C.dup_ptr[new2old[q]]);
        S.dup_src.resize(C.dup_src.size());
        parallel_chunks(nnzA, T, [&](long long qb, long long qe, int) {
            for (long long q = qb; q < qe; ++q) { const int o = new2old[q]; const int* sp = C.dup_src.data() + C.dup_ptr[o]; int* dp = S.dup_src.data() + S.dup_ptr[q];
                                                  for (int e = 0; e < C.dup_ptr[o + 1] - C.dup_ptr[o]; ++e) dp[e] = sp[e]; }
        });
    }
    lap("permuted CSC");
    S.num_sn = nsn;
    S.sn_colptr.assign(ncolptr.begin(), ncolptr.end());
    if (!finish_analysis(S, opt, lap)) return false;
    S.time_analyse = now_s() - t0;
    if (opt.verbose)
        fprintf(stderr, "[mi355x_kkt] delayed pivots: %d columns moved to their parent fronts; nsn=%d levels=%d maxfront=%d nnzL=%lld flops=%.3g  %.3fs\n",
                moved, nsn, S.num_levels, S.maxfront, (long long)S.nnz_l, (double)S.flops_factor, S.time_analyse);
    return true;
}

} // namespace mi355x

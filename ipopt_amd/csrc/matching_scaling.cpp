// matching_scaling.cpp -- symmetric matching-based scaling (the job MC64 does for HSL_MA97 / SPRAL / PARDISO; reference knobs
// ma97_scaling = mc64, spral_scaling = matching, IPARM(11)/IPARM(13): IpMa97SolverInterface.cpp:107-190,
// IpSpralSolverInterface.cpp:199-204, IpPardisoMKLSolverInterface.cpp:230-246).
//
// Restated from the published algorithm (Duff & Koster, "On algorithms for permuting large entries to the diagonal of a
// sparse matrix", SIMAX 22 (2001); symmetrisation: Duff & Pralet, SIMAX 27 (2005)): with c_ij = log max_k |a_kj| - log |a_ij| >= 0
// find a maximum-product transversal by successive shortest augmenting paths (Dijkstra on the reduced costs
// c_ij - u_i - v_j), keeping dual variables with u_i + v_j <= c_ij and equality on the matched entries.  Then
//     r_i = exp(u_i),  q_j = exp(v_j) / max_k |a_kj|,   s_i = sqrt(r_i q_i)
// gives |s_i a_ij s_j| <= 1 everywhere and = 1 on the matching.  Host code like the symbolic analysis: it runs once when the
// caller asks for it (MA97's "dynamic" policy computes it on demand and then reuses the factors).
#include "matching_scaling.h"
#include "env_knobs.h"
#include <algorithm>
#include <cmath>
#include <limits>
#include <queue>
#include <vector>
#include <chrono>
#include <thread>
#include <cstdio>
#include <cstdlib>

namespace mi355x {

bool matching_scaling(int n, const int* ptr, const int* idx, const double* absval, double* scale, int* num_unmatched)
{
    // full symmetric pattern: column j holds the rows idx[ptr[j] .. ptr[j+1]) with |values| absval[...]
    const double INF = std::numeric_limits<double>::infinity();
    const bool timing = knob_trace("matching");
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    auto lap = [&](const char* w) { if (timing) { const double t = now(); fprintf(stderr, "[match] %-12s %.3f s\n", w, t - t0); t0 = t; } };
    std::vector<double> cmaxlog(n, 0.0), u(n, INF), v(n, 0.0);
    std::vector<double> cost(ptr[n]);
    std::vector<char> empty_col(n, 0);
    {   // the logarithms are most of the set-up: columns are independent, a few threads take them
        const int T = (ptr[n] > (1 << 20)) ? (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())) : 1;
        auto work = [&](int jb, int je) {
            for (int j = jb; j < je; ++j) {
                double mx = 0.0;
                for (int p = ptr[j]; p < ptr[j + 1]; ++p) mx = std::max(mx, absval[p]);
                if (!(mx > 0.0)) { empty_col[j] = 1; continue; }
                cmaxlog[j] = std::log(mx);
                for (int p = ptr[j]; p < ptr[j + 1]; ++p) cost[p] = absval[p] > 0.0 ? cmaxlog[j] - std::log(absval[p]) : INF;
            }
        };
        if (T <= 1) work(0, n);
        else { std::vector<std::thread> th; for (int t = 0; t < T; ++t) th.emplace_back(work, (int)((long long)n * t / T), (int)((long long)n * (t + 1) / T)); for (auto& x : th) x.join(); }
    }
    lap("costs");
    // initial duals (the start MC64 makes): u_i = min_j c_ij, then v_j = min_i (c_ij - u_i) -- feasible, and every column has a tight entry;
    // greedy matching on tight entries, then one alternation per column still free: a tight row of it that is matched to a column with another
    // tight, free row gives way.  What is left goes through the shortest augmenting paths below.
    for (int j = 0; j < n; ++j) for (int p = ptr[j]; p < ptr[j + 1]; ++p) u[idx[p]] = std::min(u[idx[p]], cost[p]);
    for (int i = 0; i < n; ++i) if (u[i] == INF) u[i] = 0.0;
    for (int j = 0; j < n; ++j) {
        if (empty_col[j]) continue;
        double mn = INF;
        for (int p = ptr[j]; p < ptr[j + 1]; ++p) mn = std::min(mn, cost[p] - u[idx[p]]);
        v[j] = (mn == INF) ? 0.0 : mn;
    }
    std::vector<int> mrow(n, -1), mcol(n, -1);      // mrow[i] = column matched to row i, mcol[j] = row matched to column j
    auto tight = [&](int p, int j) { return cost[p] - u[idx[p]] - v[j] <= 0.0; };
    for (int j = 0; j < n; ++j) {
        if (empty_col[j]) continue;
        for (int p = ptr[j]; p < ptr[j + 1]; ++p) {
            const int i = idx[p];
            if (mrow[i] < 0 && tight(p, j)) { mrow[i] = j; mcol[j] = i; break; }
        }
    }
    for (int j = 0; j < n; ++j) {
        if (empty_col[j] || mcol[j] >= 0) continue;
        for (int p = ptr[j]; p < ptr[j + 1] && mcol[j] < 0; ++p) {
            const int i = idx[p];
            if (!tight(p, j)) continue;
            const int j2 = mrow[i];                 // (matched: the greedy pass would have taken a free tight row)
            if (j2 < 0) { mrow[i] = j; mcol[j] = i; break; }
            for (int q = ptr[j2]; q < ptr[j2 + 1]; ++q) {
                const int i2 = idx[q];
                if (mrow[i2] < 0 && tight(q, j2)) { mrow[i2] = j2; mcol[j2] = i2; mrow[i] = j; mcol[j] = i; break; }
            }
        }
    }
    { int nm = 0; for (int j = 0; j < n; ++j) nm += mcol[j] >= 0; if (timing) fprintf(stderr, "[match] greedy matched %d of %d\n", nm, n); }
    lap("greedy");
    std::vector<double> dist(n, INF);
    std::vector<int> pred(n, -1), touched, fin_rows;
    std::vector<char> done(n, 0);
    typedef std::pair<double, int> QE;
    int unmatched = 0;
    for (int j0 = 0; j0 < n; ++j0) {
        if (mcol[j0] >= 0 || empty_col[j0]) { if (empty_col[j0]) ++unmatched; continue; }
        std::priority_queue<QE, std::vector<QE>, std::greater<QE>> heap;
        touched.clear(); fin_rows.clear();
        int j = j0, ifree = -1;
        double base = 0.0, delta = INF;
        while (true) {
            for (int p = ptr[j]; p < ptr[j + 1]; ++p) {
                const int i = idx[p];
                if (done[i] || cost[p] == INF) continue;
                const double nd = base + (cost[p] - u[i] - v[j]);
                if (nd < dist[i]) { if (dist[i] == INF) touched.push_back(i); dist[i] = nd; pred[i] = j; heap.push(QE(nd, i)); }
            }
            int inext = -1;
            while (!heap.empty()) { const QE t = heap.top(); heap.pop(); if (!done[t.second] && t.first <= dist[t.second]) { inext = t.second; break; } }
            if (inext < 0) break;                       // no augmenting path: structurally deficient
            done[inext] = 1; fin_rows.push_back(inext);
            if (mrow[inext] < 0) { ifree = inext; delta = dist[inext]; break; }
            j = mrow[inext]; base = dist[inext];
        }
        if (ifree >= 0) {
            // dual update (finalised rows / their columns, the root column), then augment along the predecessors
            for (int i : fin_rows) {
                const double d = delta - dist[i];
                u[i] -= d;
                if (mrow[i] >= 0) v[mrow[i]] += d;
            }
            v[j0] += delta;
            int i = ifree;
            while (true) {
                const int jc = pred[i], inxt = mcol[jc];
                mrow[i] = jc; mcol[jc] = i;
                if (jc == j0) break;
                i = inxt;
            }
        } else ++unmatched;
        for (int i : touched) { dist[i] = INF; pred[i] = -1; done[i] = 0; }
    }
    lap("augment");
    for (int i = 0; i < n; ++i) {
        const double r = std::exp(u[i]), q = empty_col[i] ? 1.0 : std::exp(v[i] - cmaxlog[i]);
        double s = std::sqrt(r * q);
        if (!(s > 0.0) || !std::isfinite(s)) s = 1.0;
        scale[i] = s;
    }
    if (num_unmatched) *num_unmatched = unmatched;
    return true;
}

} // namespace mi355x

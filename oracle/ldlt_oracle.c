/*
 * ldlt_oracle.c -- CPU ORACLE (test infrastructure only; never linked into, imported by or
 * executed from the product path).
 *
 * What it restates.  The arithmetic of Ipopt's KKT hot path -- sparse symmetric-indefinite
 * LDL^T with 1x1/2x2 threshold pivoting, inertia, forward/diagonal/backward solves -- is NOT in
 * /root/reference: every backend is a third-party library behind a thin adapter (SURVEY F2;
 * .coin-or/Dependencies:1-3 pins ThirdParty-Mumps stable/3.0 = MUMPS 5.x, ThirdParty-HSL
 * stable/2.2 = MA27/MA57/HSL_MA97 2.8.0, SPRAL >= v2023.03.29).  This file restates the published
 * algorithm those libraries share -- Duff & Reid, "The multifrontal solution of indefinite sparse
 * symmetric linear systems", ACM TOMS 9 (1983) (MA27); Duff, "MA57", ACM TOMS 30 (2004) -- in its
 * plainest form: symmetric Gaussian elimination on the sparse active submatrix, pivots taken in
 * minimum-degree order subject to the threshold tests
 *     1x1:  |a_pp| >= u * max_{k != p} |a_kp|
 *     2x2:  |E^{-1}| * (gamma_p, gamma_q)^T <= 1/u  componentwise,  E = [a_pp a_qp; a_qp a_qq]
 * and a candidate that fails is passed over until its row has been updated (the "delayed pivot"
 * of the multifrontal codes).  The default u = 1e-8 is the value the reference's adapters register
 * (ma27_pivtol / ma57_pivtol / ma97_u: IpMa27TSolverInterface.cpp:92-105,
 * IpMa57TSolverInterface.cpp:201-212, IpMa97SolverInterface.cpp:93-106).
 *
 * Behavioural contract followed (the reference's call sites):
 *   - triplet input, 1-based, entries in either triangle, duplicates summed
 *     (IpMumpsSolverInterface.hpp:74-77, IpTripletToCSRConverter.cpp:352-359);
 *   - number of negative eigenvalues reported like INFOG(12) / info.num_neg
 *     (IpMumpsSolverInterface.cpp:515, IpMa97SolverInterface.cpp:779);
 *   - a structurally/numerically zero pivot makes the matrix SINGULAR
 *     (IpMa27TSolverInterface.cpp:605-612, IpMa97SolverInterface.cpp:719-724);
 *   - the solution overwrites the right-hand side, rhs[irhs*n + i] (IpSparseSymLinearSolverInterface.hpp:190).
 *
 * Pinning: tests/test_oracle.py checks this oracle against (a) every boundary call recorded from the
 * reference itself (reference Ipopt + its PardisoMKLSolverInterface, tests/golden/ recordings: inertia and
 * solutions), (b) LAPACK eigvalsh / solve on dense copies, (c) by-construction inertia.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int n;
    int **col; double **val; int *len, *cap;   /* active symmetric rows (off-diagonal part) */
    double *diag;
    char *dead;
    /* degree buckets */
    int *head, *next, *prev, *deg;
    /* factors, in elimination order */
    int nsteps; int *step_p, *step_q;           /* q = -1 for a 1x1 step */
    double *d11, *d21, *d22;
    long *lptr; int *lidx; double *lv0, *lv1; long lcap, lnz;
    int num_neg, num_zero, num_two;
} Oracle;

static void row_push(Oracle* O, int i, int j, double v)
{
    if (O->len[i] == O->cap[i]) {
        O->cap[i] = O->cap[i] ? 2 * O->cap[i] : 8;
        O->col[i] = (int*)realloc(O->col[i], sizeof(int) * O->cap[i]);
        O->val[i] = (double*)realloc(O->val[i], sizeof(double) * O->cap[i]);
    }
    O->col[i][O->len[i]] = j; O->val[i][O->len[i]] = v; O->len[i]++;
}
static void bucket_remove(Oracle* O, int i)
{
    int d = O->deg[i];
    if (O->prev[i] >= 0) O->next[O->prev[i]] = O->next[i]; else O->head[d] = O->next[i];
    if (O->next[i] >= 0) O->prev[O->next[i]] = O->prev[i];
}
static void bucket_insert(Oracle* O, int i, int d)
{
    O->deg[i] = d; O->prev[i] = -1; O->next[i] = O->head[d];
    if (O->head[d] >= 0) O->prev[O->head[d]] = i;
    O->head[d] = i;
}
static void l_reserve(Oracle* O, long extra)
{
    if (O->lnz + extra > O->lcap) {
        O->lcap = 2 * (O->lnz + extra) + 1024;
        O->lidx = (int*)realloc(O->lidx, sizeof(int) * O->lcap);
        O->lv0 = (double*)realloc(O->lv0, sizeof(double) * O->lcap);
        O->lv1 = (double*)realloc(O->lv1, sizeof(double) * O->lcap);
    }
}
static void row_remove_entry(Oracle* O, int i, int j)
{
    int k, L = O->len[i];
    for (k = 0; k < L; ++k) if (O->col[i][k] == j) { O->col[i][k] = O->col[i][L - 1]; O->val[i][k] = O->val[i][L - 1]; O->len[i] = L - 1; return; }
}
static double row_absmax(const Oracle* O, int i, int skip, int* arg)
{
    double m = -1.0; int k, a = -1;
    for (k = 0; k < O->len[i]; ++k) {
        int j = O->col[i][k]; double v;
        if (j == skip) continue;
        v = fabs(O->val[i][k]);
        if (v > m) { m = v; a = j; }
    }
    if (arg) *arg = a;
    return m < 0.0 ? 0.0 : m;
}
static double row_get(const Oracle* O, int i, int j)
{
    int k; for (k = 0; k < O->len[i]; ++k) if (O->col[i][k] == j) return O->val[i][k];
    return 0.0;
}

/* eliminate pivot (p) or (p,q); pos/w0/w1 are n-sized scratch (pos initialised to -1) */
static void eliminate(Oracle* O, int p, int q, int* pos, double* w0, double* w1, int* nb, char* flag)
{
    int nnb = 0, k, t;
    double a = O->diag[p], b = 0.0, c = 0.0, det = 0.0;
    int s = O->nsteps;
    /* neighbour set and the pivot columns w0 (= column p), w1 (= column q) */
    for (k = 0; k < O->len[p]; ++k) { int j = O->col[p][k]; if (j == q) continue; if (pos[j] < 0) { pos[j] = nnb; nb[nnb] = j; w0[nnb] = 0; w1[nnb] = 0; nnb++; } w0[pos[j]] = O->val[p][k]; }
    if (q >= 0) {
        b = row_get(O, p, q); c = O->diag[q]; det = a * c - b * b;
        for (k = 0; k < O->len[q]; ++k) { int j = O->col[q][k]; if (j == p) continue; if (pos[j] < 0) { pos[j] = nnb; nb[nnb] = j; w0[nnb] = 0; w1[nnb] = 0; nnb++; } w1[pos[j]] = O->val[q][k]; }
    }
    O->step_p[s] = p; O->step_q[s] = q; O->d11[s] = a; O->d21[s] = b; O->d22[s] = c; O->lptr[s] = O->lnz;
    l_reserve(O, nnb);
    for (t = 0; t < nnb; ++t) {
        double l0, l1 = 0.0;
        if (q < 0) l0 = w0[t] / a;
        else { l0 = (c * w0[t] - b * w1[t]) / det; l1 = (a * w1[t] - b * w0[t]) / det; }
        O->lidx[O->lnz] = nb[t]; O->lv0[O->lnz] = l0; O->lv1[O->lnz] = l1; O->lnz++;
    }
    O->lptr[s + 1] = O->lnz; O->nsteps = s + 1;
    if (q < 0) { if (a < 0) O->num_neg++; }
    else { O->num_two++; if (det < 0) O->num_neg += 1; else if (a + c < 0) O->num_neg += 2; }
    /* the neighbour marks in pos[] are reused below as "index into nb"; a second scratch marks columns of row i */
    for (t = 0; t < nnb; ++t) {
        int i = nb[t]; long base = O->lptr[s];
        double l0 = O->lv0[base + t], l1 = O->lv1[base + t];
        int L, u_;
        /* drop the pivot entries from row i */
        row_remove_entry(O, i, p); if (q >= 0) row_remove_entry(O, i, q);
        /* a_ii */
        O->diag[i] -= l0 * w0[t] + l1 * w1[t];
        /* existing entries a_ij, j a neighbour: update in place and tag them */
        L = O->len[i];
        for (k = 0; k < L; ++k) { int j = O->col[i][k]; int pj = pos[j]; if (pj >= 0) { O->val[i][k] -= l0 * w0[pj] + l1 * w1[pj]; } }
        /* fill-in: neighbours j not yet present in row i.  Mark present ones through a sign trick on nb-local flags */
        {
            memset(flag, 0, nnb);
            for (k = 0; k < L; ++k) { int pj = pos[O->col[i][k]]; if (pj >= 0) flag[pj] = 1; }
            for (u_ = 0; u_ < nnb; ++u_) if (u_ != t && !flag[u_]) { double v = -(l0 * w0[u_] + l1 * w1[u_]); row_push(O, i, nb[u_], v); }
        }
        bucket_remove(O, i); bucket_insert(O, i, O->len[i]);
    }
    for (t = 0; t < nnb; ++t) pos[nb[t]] = -1;
    O->dead[p] = 1; bucket_remove(O, p); O->len[p] = 0;
    if (q >= 0) { O->dead[q] = 1; bucket_remove(O, q); O->len[q] = 0; }
}

/*
 * Factor + solve.  irn/jcn: triplet (base = 0 or 1), a: values, rhs: nrhs columns of length n,
 * overwritten by the solution.  Returns 0, or 1 if the matrix is singular (num_zero > 0; the
 * solution is then computed with the zero pivots' unknowns set to 0).
 */
int kkt_oracle_factor_solve(int n, int nnz, const int* irn, const int* jcn, const double* a, int base,
                            double u, double small, int nrhs, double* rhs,
                            int* num_neg, int* num_zero, int* num_two)
{
    Oracle O; int i, k, t, s, d, dmax;
    int *pos, *nb; double *w0, *w1; char* flag; int dstart = 0;
    memset(&O, 0, sizeof(O));
    O.n = n;
    O.col = (int**)calloc(n + 1, sizeof(int*)); O.val = (double**)calloc(n + 1, sizeof(double*));
    O.len = (int*)calloc(n + 1, sizeof(int)); O.cap = (int*)calloc(n + 1, sizeof(int));
    O.diag = (double*)calloc(n + 1, sizeof(double)); O.dead = (char*)calloc(n + 1, 1);
    O.head = (int*)malloc(sizeof(int) * (n + 2)); O.next = (int*)malloc(sizeof(int) * (n + 1));
    O.prev = (int*)malloc(sizeof(int) * (n + 1)); O.deg = (int*)calloc(n + 1, sizeof(int));
    O.step_p = (int*)malloc(sizeof(int) * (n + 1)); O.step_q = (int*)malloc(sizeof(int) * (n + 1));
    O.d11 = (double*)malloc(sizeof(double) * (n + 1)); O.d21 = (double*)malloc(sizeof(double) * (n + 1)); O.d22 = (double*)malloc(sizeof(double) * (n + 1));
    O.lptr = (long*)malloc(sizeof(long) * (n + 2));
    pos = (int*)malloc(sizeof(int) * (n + 1)); nb = (int*)malloc(sizeof(int) * (n + 1));
    w0 = (double*)malloc(sizeof(double) * (n + 1)); w1 = (double*)malloc(sizeof(double) * (n + 1)); flag = (char*)malloc(n + 1);
    for (i = 0; i < n; ++i) pos[i] = -1;
    /* assemble, summing duplicates (sort-free: per-row linear search is fine for KKT row lengths,
       but to stay O(nnz) on big inputs use the pos[] scatter per row after bucketing by row) */
    {
        int* cnt = (int*)calloc(n + 1, sizeof(int)); int* start; int* ord; int* lo = (int*)malloc(sizeof(int) * (nnz + 1)); int* hi = (int*)malloc(sizeof(int) * (nnz + 1));
        for (t = 0; t < nnz; ++t) { int r = irn[t] - base, c = jcn[t] - base; lo[t] = r < c ? r : c; hi[t] = r < c ? c : r; if (lo[t] != hi[t]) cnt[lo[t]]++; }
        start = (int*)malloc(sizeof(int) * (n + 1)); start[0] = 0; for (i = 0; i < n; ++i) start[i + 1] = start[i] + cnt[i];
        ord = (int*)malloc(sizeof(int) * (nnz + 1)); memset(cnt, 0, sizeof(int) * (n + 1));
        for (t = 0; t < nnz; ++t) { if (lo[t] == hi[t]) O.diag[lo[t]] += a[t]; else ord[start[lo[t]] + cnt[lo[t]]++] = t; }
        for (i = 0; i < n; ++i) {
            /* distinct columns of the strictly-lower entries whose smaller index is i */
            int first = O.len[i];
            for (k = start[i]; k < start[i + 1]; ++k) { t = ord[k]; if (pos[hi[t]] < 0) { pos[hi[t]] = O.len[i]; row_push(&O, i, hi[t], a[t]); } else O.val[i][pos[hi[t]]] += a[t]; }
            for (k = first; k < O.len[i]; ++k) pos[O.col[i][k]] = -1;
        }
        /* mirror: entry (i,j) with i<j stored in row i so far; add to row j */
        { int* len0 = (int*)malloc(sizeof(int) * (n + 1)); for (i = 0; i < n; ++i) len0[i] = O.len[i];
          for (i = 0; i < n; ++i) for (k = 0; k < len0[i]; ++k) if (O.col[i][k] > i) row_push(&O, O.col[i][k], i, O.val[i][k]);
          free(len0); }
        free(cnt); free(start); free(ord); free(lo); free(hi);
    }
    for (d = 0; d <= n; ++d) O.head[d] = -1;
    for (i = n - 1; i >= 0; --i) bucket_insert(&O, i, O.len[i]);

    /* elimination */
    {
        int remaining = n;
        while (remaining > 0) {
            int chosen = -1, partner = -1; 
            int best_fallback = -1; double best_ratio = -1.0;
            int lowest = -1;
            dmax = n;
            for (d = dstart; d <= dmax && chosen < 0; ++d) {
                if (O.head[d] >= 0 && lowest < 0) lowest = d;
                for (i = O.head[d]; i >= 0 && chosen < 0; i = O.next[i]) {
                    int j = -1; double g = row_absmax(&O, i, -1, &j), aii = fabs(O.diag[i]);
                    if (g == 0.0) { chosen = i; partner = -1; break; }        /* isolated: pivot (or zero pivot) */
                    if (aii > small && aii >= u * g) { chosen = i; partner = -1; break; }
                    /* 2x2 candidate with the largest off-diagonal */
                    if (j >= 0) {
                        double b = row_get(&O, i, j), c = O.diag[j], a_ = O.diag[i], det = a_ * c - b * b;
                        double gi = row_absmax(&O, i, j, 0), gj = row_absmax(&O, j, i, 0);
                        if (fabs(det) > small * small && fabs(det) >= 0.0) {
                            double t1 = (fabs(c) * gi + fabs(b) * gj), t2 = (fabs(b) * gi + fabs(a_) * gj);
                            if (t1 * u <= fabs(det) && t2 * u <= fabs(det)) { chosen = i; partner = j; break; }
                        }
                    }
                    if (aii > small) { double r = aii / g; if (r > best_ratio) { best_ratio = r; best_fallback = i; } }
                }
            }
            if (chosen < 0) {   /* nothing passes the threshold anywhere: relax (this is where MA27 would enlarge the front) */
                if (best_fallback >= 0) { chosen = best_fallback; partner = -1; }
                else { for (d = 0; d <= n && chosen < 0; ++d) if (O.head[d] >= 0) chosen = O.head[d]; partner = -1; }
            }
            dstart = (lowest < 0 ? 0 : lowest) - 2; if (dstart < 0) dstart = 0;
            if (partner < 0 && fabs(O.diag[chosen]) <= small) {
                /* zero pivot: singular; eliminate with an infinite pivot (unknown := 0, no update) */
                int s2 = O.nsteps; O.num_zero++;
                for (k = 0; k < O.len[chosen]; ++k) { int j = O.col[chosen][k]; row_remove_entry(&O, j, chosen); bucket_remove(&O, j); bucket_insert(&O, j, O.len[j]); }
                O.step_p[s2] = chosen; O.step_q[s2] = -2; O.d11[s2] = 0; O.d21[s2] = 0; O.d22[s2] = 0; O.lptr[s2] = O.lnz; O.lptr[s2 + 1] = O.lnz; O.nsteps++;
                O.dead[chosen] = 1; bucket_remove(&O, chosen); O.len[chosen] = 0; remaining -= 1;
                continue;
            }
            eliminate(&O, chosen, partner, pos, w0, w1, nb, flag);
            remaining -= (partner >= 0) ? 2 : 1;
        }
    }
    /* solves */
    for (t = 0; t < nrhs; ++t) {
        double* x = rhs + (size_t)t * n; long q_;
        for (s = 0; s < O.nsteps; ++s) {
            int p = O.step_p[s], q = O.step_q[s];
            if (q == -2) { x[p] = 0.0; continue; }
            if (q < 0) { double xp = x[p]; for (q_ = O.lptr[s]; q_ < O.lptr[s + 1]; ++q_) x[O.lidx[q_]] -= O.lv0[q_] * xp; }
            else { double xp = x[p], xq = x[q]; for (q_ = O.lptr[s]; q_ < O.lptr[s + 1]; ++q_) x[O.lidx[q_]] -= O.lv0[q_] * xp + O.lv1[q_] * xq; }
        }
        for (s = 0; s < O.nsteps; ++s) {
            int p = O.step_p[s], q = O.step_q[s];
            if (q == -2) continue;
            if (q < 0) x[p] /= O.d11[s];
            else { double a_ = O.d11[s], b = O.d21[s], c = O.d22[s], det = a_ * c - b * b, xp = x[p], xq = x[q]; x[p] = (c * xp - b * xq) / det; x[q] = (a_ * xq - b * xp) / det; }
        }
        for (s = O.nsteps - 1; s >= 0; --s) {
            int p = O.step_p[s], q = O.step_q[s];
            if (q == -2) continue;
            if (q < 0) { double acc = 0; for (q_ = O.lptr[s]; q_ < O.lptr[s + 1]; ++q_) acc += O.lv0[q_] * x[O.lidx[q_]]; x[p] -= acc; }
            else { double a0 = 0, a1 = 0; for (q_ = O.lptr[s]; q_ < O.lptr[s + 1]; ++q_) { a0 += O.lv0[q_] * x[O.lidx[q_]]; a1 += O.lv1[q_] * x[O.lidx[q_]]; } x[p] -= a0; x[q] -= a1; }
        }
    }
    if (num_neg) *num_neg = O.num_neg;
    if (num_zero) *num_zero = O.num_zero;
    if (num_two) *num_two = O.num_two;
    for (i = 0; i < n; ++i) { free(O.col[i]); free(O.val[i]); }
    free(O.col); free(O.val); free(O.len); free(O.cap); free(O.diag); free(O.dead); free(O.head); free(O.next); free(O.prev); free(O.deg);
    free(O.step_p); free(O.step_q); free(O.d11); free(O.d21); free(O.d22); free(O.lptr); free(O.lidx); free(O.lv0); free(O.lv1);
    free(pos); free(nb); free(w0); free(w1); free(flag);
    return O.num_zero > 0 ? 1 : 0;
}

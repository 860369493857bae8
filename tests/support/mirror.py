"""numpy 'block multifrontal' walk over the symbolic structures exported by the C ABI
(mi355x_kkt_get_symbolic).  TEST SUPPORT ONLY: it validates the host-side symbolic analysis
(permutation, supernode row lists, child->parent relative indices, A scatter map, level schedule)
on machines without a GPU by eliminating each front's pivot block as one dense block.  It is not a
product code path and is never imported by ipopt_amd."""
from __future__ import annotations

import numpy as np


def fetch(solver):
    I = solver.info()
    g = solver.symbolic
    return dict(info=I, perm=g(0, I.n), colptr=g(1, I.num_sn + 1), rowptr=g(2, I.num_sn + 1), rows=g(3, I.sum_sn_rows),
                parent=g(4, I.num_sn), level=g(5, I.num_sn), rel=g(6, I.sum_sn_rows), acolptr=g(7, I.n + 1),
                arow=g(8, I.nnz_a), t2s=g(9, I.nnz_in), pair=g(10, I.n), owner=g(11, I.num_sn), apos=g(12, I.nnz_a),
                glo=g(18, I.num_sn), gsz=g(19, I.num_sn), gdepth=g(20, I.num_sn), cls=g(23, I.num_sn))


def factor_solve(sym, vals, rhs, only=None):
    """returns (x, num_neg) using the symbolic structures; dense block elimination per front."""
    I = sym["info"]
    n, nsn = I.n, I.num_sn
    aval = np.zeros(I.nnz_a)
    np.add.at(aval, sym["t2s"], vals)
    colptr, rowptr, rows, rel, parent = sym["colptr"], sym["rowptr"], sym["rows"], sym["rel"], sym["parent"]
    children = [[] for _ in range(nsn)]
    for s in range(nsn):
        if parent[s] >= 0:
            assert parent[s] > s
            children[parent[s]].append(s)
    F11i, F21, cbs, negs = [None] * nsn, [None] * nsn, [None] * nsn, 0
    b = rhs[sym["perm"]].astype(float).copy()
    cvec = [None] * nsn
    for s in range(nsn):
        c0, c1 = colptr[s], colptr[s + 1]
        k = c1 - c0
        r = rows[rowptr[s]:rowptr[s + 1]]
        m = r.shape[0]
        assert np.all(r[:k] == np.arange(c0, c1)) and np.all(np.diff(r[k:]) > 0) and (m == k or r[k] >= c1)
        F = np.zeros((m, m))
        q0, q1 = sym["acolptr"][c0], sym["acolptr"][c1]
        pos = sym["apos"][q0:q1]
        li, lj = pos % m, pos // m
        # cross-check the scatter map against the row lists
        assert np.all(r[li] == sym["arow"][q0:q1]) and np.all(lj < k) and np.all(li >= lj)
        F[li, lj] += aval[q0:q1]
        F[lj, li] = F[li, lj]
        bs = np.zeros(m); bs[:k] = b[c0:c1]
        for ch in children[s]:
            kc = colptr[ch + 1] - colptr[ch]
            rl = rel[rowptr[ch] + kc:rowptr[ch + 1]]
            assert np.all(rl >= 0) and np.all(np.diff(rl) > 0) and np.all(r[rl] == rows[rowptr[ch] + kc:rowptr[ch + 1]])
            F[np.ix_(rl, rl)] += cbs[ch]
            bs[rl] += cvec[ch]
            cbs[ch] = None
        A11, A21, A22 = F[:k, :k], F[k:, :k], F[k:, k:]
        w = np.linalg.eigvalsh(A11)
        negs += int((w < 0).sum())
        inv = np.linalg.inv(A11)
        F11i[s], F21[s] = inv, A21.copy()
        cbs[s] = A22 - A21 @ inv @ A21.T
        y = bs[:k]
        cvec[s] = bs[k:] - A21 @ (inv @ y)
        b[c0:c1] = y
    x = np.zeros(n)
    for s in range(nsn - 1, -1, -1):
        c0, c1 = colptr[s], colptr[s + 1]
        k = c1 - c0
        r = rows[rowptr[s]:rowptr[s + 1]]
        x[c0:c1] = F11i[s] @ (b[c0:c1] - F21[s].T @ x[r[k:]])
    out = np.zeros(n)
    out[sym["perm"]] = x
    return out, negs


# ------------------------------------------------------------------------------------------------------
# Executable SPECIFICATION of the pivoting rules of ipopt_amd/csrc/kernels_fronts.hip.inc / kernels_big.hip.inc (ldlt_reg / k_big_trsm; parts of the numeric.hip translation unit), in
# plain numpy: same candidate order, same Bunch-Kaufman preference, same MA27/MA57 threshold tests against
# the whole front column, same pass-over / forced-pivot / zero-pivot rules.  The CPU tests pin it against
# the oracle (inertia, solution) at several u; the GPU tests compare the HIP kernels' pivot statistics
# (num_two, num_delay, u_sensitive) with it.  TEST SUPPORT ONLY.
# ------------------------------------------------------------------------------------------------------
BK_ALPHA = 0.6403882032022076
BK_ALPHA0 = 0.1     # |a_jj| >= BK_ALPHA0 * (whole remaining column): 1x1 at j without a partner search
PIV_PERT = 1e-10
ZERO_REL = 1e-14


def ldlt_front(F, k, u, u2, small=1e-20, see_update_rows=True, cnorm=None):
    """In-place LDL^T of the first k (fully-summed) rows/columns of the symmetric m x m front F.
    Returns dict(ord, ptype, dinv, doff, L (m x k, physical rows, column = elimination step), nneg, nzero, ntwo, ndelay, chg)."""
    m = F.shape[0]
    cm0 = np.abs(F[:, :k]).max(axis=0) if (k and F.size) else np.zeros(k)      # scale of each fully-summed column as assembled ...
    if cnorm is not None:
        cm0 = np.maximum(cm0, cnorm)                                           # ... or in the input matrix, whichever is larger
    alive = list(range(k))
    tryb = list(alive)
    force = False
    upd = list(range(k, m)) if see_update_rows else []
    L = np.zeros((m, k)); order = []; ptype = []; dinv = []; doff = []
    st = dict(nneg=0, nzero=0, ntwo=0, ndelay=0, chg=0, delayed=set())      # delayed: physical rows that were alive when every candidate had failed

    def colmax(col, rows):
        return max((abs(F[i, col]) for i in rows), default=0.0)

    while alive:
        if not tryb:
            force = True; tryb = list(alive)
            st["delayed"].update(alive)          # every candidate failed: they are the DELAYED pivots (the kernels mark them; the host moves them to the parent front)
        j = tryb[0]
        ztol = max(small, ZERO_REL * cm0[j])
        ajj = abs(F[j, j])
        fs = [i for i in alive if i != j]
        gj = max(colmax(j, fs), colmax(j, upd))
        lam = colmax(j, fs) if fs else -1.0
        bkneed = gj * BK_ALPHA0 > ajj
        thfail = gj * u > ajj
        if not force and gj * u2 > ajj:
            st["chg"] = 1
        p, q, zero = j, -1, False
        if bkneed or thfail or not (ajj > ztol):
            uu = 0.0 if force else u
            sel = -1
            r = -1
            if lam > 0.0:
                r = next(i for i in fs if abs(F[i, j]) == lam)
                fsr = [i for i in alive if i != r]
                sig = colmax(r, fsr)
                gr = max(sig, colmax(r, upd))
                rest = [i for i in alive if i not in (j, r)] + upd
                gj2, gr2 = colmax(j, rest), colmax(r, rest)
                a, b, c = F[j, j], F[r, j], F[r, r]
                arr, ab = abs(c), abs(b)
                det = a * c - b * b; adet = abs(det)
                pref = 0 if (ajj >= BK_ALPHA * lam or ajj * sig >= BK_ALPHA * lam * lam) else (1 if arr >= BK_ALPHA * sig else 2)
                t1, t2 = arr * gj2 + ab * gr2, ab * gj2 + ajj * gr2
                ztr = max(small, ZERO_REL * cm0[r])
                nz2 = adet > max(small, ZERO_REL * max(ajj * arr, ab * ab))
                ok = [ajj > ztol and ajj >= uu * gj, arr > ztr and arr >= uu * gr, nz2 and t1 * uu <= adet and t2 * uu <= adet]
                if ok[pref]:
                    sel = pref
                elif ok[0]:
                    sel = 0
                elif ok[2]:
                    sel = 2
                elif ok[1]:
                    sel = 1
                if sel >= 0:
                    fu = [ajj < u * gj, arr < u * gr, t1 * u > adet or t2 * u > adet][sel]
                    fu2 = [ajj < u2 * gj, arr < u2 * gr, t1 * u2 > adet or t2 * u2 > adet][sel]
                    if fu:
                        st["ndelay"] += 2 if sel == 2 else 1
                    if fu2 and not force:
                        st["chg"] = 1
            else:
                if ajj > ztol and ajj >= uu * gj:
                    sel = 0
                    if ajj < u * gj:
                        st["ndelay"] += 1
                    if ajj < u2 * gj and not force:
                        st["chg"] = 1
                elif not (ajj > ztol) and not (gj > ztol):
                    sel, zero = 0, True
            if sel < 0:
                if not force:
                    tryb.remove(j)
                    continue
                sel, zero = 0, True
            if sel == 1:
                p = r
            elif sel == 2:
                q = r
        force = False
        s = len(order)
        live = [i for i in alive if i not in (p, q)] + list(range(k, m))
        if q >= 0:
            a, b, c = F[p, p], F[q, p], F[q, q]
            det = a * c - b * b
            w0, w1 = F[:, p].copy(), F[:, q].copy()
            l0, l1 = (c * w0 - b * w1) / det, (a * w1 - b * w0) / det
            for i in live:
                L[i, s], L[i, s + 1] = l0[i], l1[i]
            idx = np.array(live, dtype=int)
            F[np.ix_(idx, idx)] -= np.outer(l0[idx], w0[idx]) + np.outer(l1[idx], w1[idx])
            order += [p, q]; ptype += [2, 3]; dinv += [c / det, a / det]; doff += [-b / det, 0.0]
            st["nneg"] += 1 if det < 0 else (2 if a + c < 0 else 0)
            st["ntwo"] += 1
            alive.remove(p); alive.remove(q)
        else:
            d = F[p, p]
            if zero:
                st["nzero"] += 1
                d = -PIV_PERT if d < 0 else PIV_PERT
            w0 = F[:, p].copy()
            l0 = w0 / d
            for i in live:
                L[i, s] = l0[i]
            idx = np.array(live, dtype=int)
            F[np.ix_(idx, idx)] -= np.outer(l0[idx], w0[idx])
            order += [p]; ptype += [1]; dinv += [1.0 / d]; doff += [0.0]
            if d < 0:
                st["nneg"] += 1
            alive.remove(p)
        tryb = list(alive)
    st.update(ord=np.array(order, dtype=int), ptype=ptype, dinv=np.array(dinv), doff=np.array(doff), L=L)
    return st


BIG_FRONT = 128      # fronts above this order take the blocked path: in-block test + a posteriori test on the rows below


def is_big(sym, s, m):
    """The blocked path: class 3 in the library's own classification (symbolic.cpp: order > 128)."""
    return sym["cls"][s] == 3 if "cls" in sym else m > BIG_FRONT

FAST_U = 1e-4        # a pivot block taken in natural order is accepted iff every multiplier is <= 1 / max(u, u2, FAST_U)


def ldlt_block_static(A, k, u, u2, small=1e-20, cnorm=None):
    """The fast path of a big front's pivot block (kernels_fronts.hip.inc: ldlt_blocked_static): the k x k block is eliminated in NATURAL
    order with 1x1 pivots, nothing is decided per pivot, and the result is accepted A POSTERIORI iff every pivot is clear of the
    zero threshold of the block and every multiplier inside the block is <= 1 / max(u, u2, FAST_U).  Returns the same dict as
    ldlt_front, or None when the block is rejected (the caller then runs the strict rule on the untouched block)."""
    A = A.copy()
    cm = np.abs(A).max(axis=0) if k else np.zeros(0)
    if cnorm is not None:
        cm = np.maximum(cm, cnorm)
    zmax = max(small, ZERO_REL * (cm.max() if k else 0.0))
    gmax = 1.0 / max(u, u2, FAST_U)
    L = np.zeros((k, k)); dinv = np.zeros(k); nneg = 0
    for j in range(k):
        d = A[j, j]
        if not abs(d) > zmax:
            return None
        w = A[j + 1:, j].copy()
        l = w / d
        if l.size and np.abs(l).max() > gmax:
            return None
        L[j + 1:, j] = l
        A[j + 1:, j + 1:] -= np.outer(l, w)
        dinv[j] = 1.0 / d
        nneg += int(d < 0)
    return dict(ord=np.arange(k), ptype=[1] * k, dinv=dinv, doff=np.zeros(k), L=L, nneg=nneg, nzero=0, ntwo=0, ndelay=0, chg=0, delayed=set())


FAST16_MIN_M = 65    # fronts of order 65 .. 128 (the 256-thread front kernel) with <= 16 pivots try the static path first


def ldlt_front_static(F, k, u, u2, small=1e-20, cnorm=None):
    """The fast path of a front of order 65 .. 128 with k <= 16 pivots (kernels_fronts.hip.inc: front_fast16): the k fully-summed columns are
    eliminated in NATURAL order with 1x1 pivots, nothing decided per pivot, and the result is accepted A POSTERIORI iff every pivot is
    clear of the front's zero threshold and every multiplier -- update rows included: a front of this size sees its whole column -- is
    <= 1 / max(u, u2, FAST_U).  On acceptance F holds the Schur complement in F[k:, k:] and the same dict as ldlt_front is returned;
    on rejection F is untouched and None is returned (the strict rule then runs)."""
    m = F.shape[0]
    A = F.copy()
    cm = np.abs(A[:, :k]).max(axis=0) if k else np.zeros(0)
    if cnorm is not None:
        cm = np.maximum(cm, cnorm)
    zmax = max(small, ZERO_REL * (cm.max() if k else 0.0))
    gmax = 1.0 / max(u, u2, FAST_U)
    L = np.zeros((m, k)); dinv = np.zeros(k); nneg = 0
    for j in range(k):
        d = A[j, j]
        if not abs(d) > zmax:
            return None
        w = A[j + 1:, j].copy()
        l = w / d
        if l.size and np.abs(l).max() > gmax:
            return None
        L[j + 1:, j] = l
        A[j + 1:, j + 1:] -= np.outer(l, w)
        dinv[j] = 1.0 / d
        nneg += int(d < 0)
    F[:, :] = A
    return dict(ord=np.arange(k), ptype=[1] * k, dinv=dinv, doff=np.zeros(k), L=L, nneg=nneg, nzero=0, ntwo=0, ndelay=0, chg=0, delayed=set())


def factor_solve_pivoted(sym, vals, rhs, u=1e-8, u2=1e-4, small=1e-20, fast_blocks=True, debug=False):
    """multifrontal LDL^T with the pivoting rules of the HIP kernels (no scaling: use scaling=0 on the GPU side).
    Returns (x, dict(num_neg, num_zero, num_two, num_delay, u_sensitive, num_fast)); num_fast = pivot blocks of big fronts
    accepted on the natural-order a-posteriori path (ldlt_block_static), the others took the strict rule."""
    I = sym["info"]
    n, nsn = I.n, I.num_sn
    aval = np.zeros(I.nnz_a)
    np.add.at(aval, sym["t2s"], vals)
    colptr, rowptr, rows, rel, parent = sym["colptr"], sym["rowptr"], sym["rows"], sym["rel"], sym["parent"]
    children = [[] for _ in range(nsn)]
    for s in range(nsn):
        if parent[s] >= 0:
            children[parent[s]].append(s)
    fac, cbs, cvec = [None] * nsn, [None] * nsn, [None] * nsn
    tot = dict(num_neg=0, num_zero=0, num_two=0, num_delay=0, u_sensitive=0, num_fast=0, marks=[])     # marks: permuted columns flagged as delayed pivots
    if debug:
        tot["dbg"] = []                # per front: (s, c0, k, m, pivot order, pivot types, diagonal of D^{-1})
    b = rhs[sym["perm"]].astype(float).copy()
    # inf-norm of every column of the (symmetric) input matrix, permuted numbering
    cn = np.zeros(n)
    col_of = np.repeat(np.arange(n), np.diff(sym["acolptr"]))
    np.maximum.at(cn, col_of, np.abs(aval)); np.maximum.at(cn, sym["arow"], np.abs(aval))
    for s in range(nsn):
        c0, c1 = colptr[s], colptr[s + 1]
        k = c1 - c0
        r = rows[rowptr[s]:rowptr[s + 1]]
        m = r.shape[0]
        F = np.zeros((m, m))
        q0, q1 = sym["acolptr"][c0], sym["acolptr"][c1]
        pos = sym["apos"][q0:q1]
        li, lj = pos % m, pos // m
        F[li, lj] += aval[q0:q1]
        off = li != lj
        F[lj[off], li[off]] += aval[q0:q1][off]
        bs = np.zeros(m); bs[:k] = b[c0:c1]
        for ch in children[s]:
            kc = colptr[ch + 1] - colptr[ch]
            rl = rel[rowptr[ch] + kc:rowptr[ch + 1]]
            F[np.ix_(rl, rl)] += cbs[ch]
            bs[rl] += cvec[ch]
            cbs[ch] = None
        if not is_big(sym, s, m):
            # static-order path first: fronts of order <= 16 (k_front_dpp16) and of order 65 .. 128 with <= 16 pivots (front_fast16)
            st = ldlt_front_static(F, k, u, u2, small, cnorm=cn[c0:c1]) if (fast_blocks and (m <= 16 or (m >= FAST16_MIN_M and k <= 16))) else None
            if st is None:
                st = ldlt_front(F, k, u, u2, small, cnorm=cn[c0:c1])
            P = st["ord"]
            L11 = np.tril(st["L"][P, :], -1) + np.eye(k)
            L21 = st["L"][k:, :]
            cb = F[k:, k:].copy()
        else:
            A11 = F[:k, :k].copy()
            st = ldlt_block_static(A11, k, u, u2, small, cnorm=cn[c0:c1]) if (fast_blocks and k <= 64) else None
            if st is not None:
                tot["num_fast"] += 1
            else:
                st = ldlt_front(A11, k, u, u2, small, see_update_rows=False, cnorm=cn[c0:c1])
            P = st["ord"]
            L11 = np.tril(st["L"][P, :], -1) + np.eye(k)
            D = np.zeros((k, k))
            Dinv = _dinv_matrix(st)
            W = np.linalg.solve(L11, F[k:, :k][:, P].T).T            # W21 = A21 P L11^{-T}
            L21 = W @ Dinv
            if m > k:          # a posteriori: a column of L21 with a multiplier above 1/u is a failed pivot (k_big_trsm) -- the column and its 2x2 partner are marked
                bad = np.nonzero(np.abs(L21).max(axis=0) * u > 1.0)[0]
                st["ndelay"] += int(bad.size)
                for j in bad:
                    st["delayed"].add(int(P[j]))
                    if st["ptype"][j] == 2:
                        st["delayed"].add(int(P[j + 1]))
                    elif st["ptype"][j] == 3:
                        st["delayed"].add(int(P[j - 1]))
            cb = F[k:, k:] - L21 @ W.T
        tot["num_neg"] += st["nneg"]; tot["num_zero"] += st["nzero"]; tot["num_two"] += st["ntwo"]; tot["num_delay"] += st["ndelay"]
        tot["u_sensitive"] |= st["chg"]
        tot["marks"] += [c0 + int(p) for p in sorted(st["delayed"])]
        Dinv = _dinv_matrix(st)
        fac[s] = (P, L11, L21, Dinv)
        if "dbg" in tot:
            tot["dbg"].append((s, c0, k, m, P.copy(), np.array(st["ptype"]), np.diag(Dinv).copy()))
        cbs[s] = cb
        y = np.linalg.solve(L11, bs[:k][P])
        cvec[s] = bs[k:] - L21 @ y
        b[c0:c1] = Dinv @ y                       # z, in pivot order
    x = np.zeros(n)
    for s in range(nsn - 1, -1, -1):
        c0, c1 = colptr[s], colptr[s + 1]
        k = c1 - c0
        r = rows[rowptr[s]:rowptr[s + 1]]
        P, L11, L21, _ = fac[s]
        xp = np.linalg.solve(L11.T, b[c0:c1] - L21.T @ x[r[k:]])
        x[c0 + P] = xp
    out = np.zeros(n)
    out[sym["perm"]] = x
    return out, tot


def _dinv_matrix(st):
    k = len(st["ptype"])
    Di = np.zeros((k, k))
    for j, pt in enumerate(st["ptype"]):
        Di[j, j] = st["dinv"][j]
        if pt == 2:
            Di[j, j + 1] = Di[j + 1, j] = st["doff"][j]
    return Di



def factor_solve_delayed(solver, vals, rhs, u=1e-8, u2=1e-4, small=1e-20, rounds=8, fast_blocks=True):
    """The delayed-pivot loop of the C library (api.cpp delay_and_refactor) with the numpy specification in place of the HIP factorisation:
    factor on the current structure; while pivots had to be forced, hand the marked columns to mi355x_kkt_delay_columns (the host-side
    structural edit, symbolic.cpp restructure_delays -- the SAME code the product runs) and factor again on the edited structure.
    Returns (x, stats of the last factorisation, number of structure edits, columns moved in total)."""
    edits = moved_total = 0
    while True:
        sym = fetch(solver)
        x, st = factor_solve_pivoted(sym, vals, rhs, u=u, u2=u2, small=small, fast_blocks=fast_blocks)
        if edits >= rounds or not st["marks"]:          # (marks without a counted failure exist: a forced candidate with nothing usable is a ZERO pivot, not a num_delay)
            return x, st, edits, moved_total
        moved = solver.delay_columns(sym["perm"][np.array(st["marks"], dtype=int)] + 1)
        if moved == 0:
            return x, st, edits, moved_total
        edits += 1; moved_total += moved

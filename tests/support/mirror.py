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
                arow=g(8, I.nnz_a), t2s=g(9, I.nnz_in), pair=g(10, I.n), owner=g(11, I.num_sn), apos=g(12, I.nnz_a))


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

"""numpy stand-in for ipopt_amd.multigpu.HipEngine -- TEST SUPPORT ONLY (CPU, gloo).  Walks the symbolic structures
exported by the C ABI exactly as the HIP multi-GPU path does: own subtrees first, own contributions to the replicated
top fronts into an arena that the driver all-reduces, replicated top factorisation, mirrored solve."""
from __future__ import annotations

import numpy as np
import torch

import ipopt_amd
from . import mirror


class MirrorEngine:
    def __init__(self, rank, nranks, **opts):
        self.rank, self.nranks = rank, nranks
        self.s = ipopt_amd.KKTSolver(nranks=nranks, rank=rank, **opts)

    def analyse(self, n, row, col, vals):
        self.s.initialize_structure(n, row, col, vals=vals)
        self.sym = mirror.fetch(self.s)
        sy = self.sym
        I = sy["info"]
        self.n, self.nsn = I.n, I.num_sn
        self.children = [[] for _ in range(self.nsn)]
        for s in range(self.nsn):
            if sy["parent"][s] >= 0:
                self.children[sy["parent"][s]].append(s)
        self.top = [s for s in range(self.nsn) if sy["owner"][s] < 0]
        # arena squares only for replicated fronts with a rank-owned child (the subtree joins), as in numeric.hip
        self.join = [s for s in self.top if any(sy["owner"][ch] >= 0 for ch in self.children[s])]
        self.aoff, self.toff, a, t = {}, {}, 0, 0
        for s in self.top:
            self.toff[s] = t; t += sy["rowptr"][s + 1] - sy["rowptr"][s]
        for s in self.join:
            m = sy["rowptr"][s + 1] - sy["rowptr"][s]
            self.aoff[s] = a; a += m * m
        self._arena = np.zeros(a); self._toprhs = np.zeros(t)

    def _front_dims(self, s):
        sy = self.sym
        c0, c1 = sy["colptr"][s], sy["colptr"][s + 1]
        r = sy["rows"][sy["rowptr"][s]:sy["rowptr"][s + 1]]
        return c0, c1 - c0, r, r.shape[0]

    def _a_entries(self, s, F):
        sy = self.sym
        c0, k, r, m = self._front_dims(s)
        q0, q1 = sy["acolptr"][c0], sy["acolptr"][c0 + k]
        pos = sy["apos"][q0:q1]
        li, lj = pos % m, pos // m
        F[li, lj] += self.aval[q0:q1]
        off = li != lj
        F[lj[off], li[off]] += self.aval[q0:q1][off]

    def _rel(self, ch):
        sy = self.sym
        kc = sy["colptr"][ch + 1] - sy["colptr"][ch]
        return sy["rel"][sy["rowptr"][ch] + kc:sy["rowptr"][ch + 1]]

    def _eliminate(self, s, F):
        c0, k, r, m = self._front_dims(s)
        A11, A21, A22 = F[:k, :k], F[k:, :k], F[k:, k:]
        self.neg[s] = int((np.linalg.eigvalsh(A11) < 0).sum())
        inv = np.linalg.inv(A11)
        self.F11i[s], self.F21[s] = inv, A21.copy()
        self.cb[s] = A22 - A21 @ inv @ A21.T

    def factor_local(self, vals):
        sy = self.sym
        vals = np.asarray(vals)
        self.aval = np.zeros(sy["info"].nnz_a); np.add.at(self.aval, sy["t2s"], vals)
        self.F11i, self.F21, self.cb, self.neg = [None] * self.nsn, [None] * self.nsn, [None] * self.nsn, [0] * self.nsn
        for s in range(self.nsn):
            if sy["owner"][s] != self.rank:
                continue
            c0, k, r, m = self._front_dims(s)
            F = np.zeros((m, m)); self._a_entries(s, F)
            for ch in self.children[s]:
                rl = self._rel(ch); F[np.ix_(rl, rl)] += self.cb[ch]
            self._eliminate(s, F)
        self._arena[:] = 0.0
        for s in self.join:
            c0, k, r, m = self._front_dims(s)
            F = np.zeros((m, m))
            for ch in self.children[s]:
                if sy["owner"][ch] == self.rank:
                    rl = self._rel(ch); F[np.ix_(rl, rl)] += self.cb[ch]
            self._arena[self.aoff[s]:self.aoff[s] + m * m] = F.ravel()

    def arena(self):
        return torch.from_numpy(self._arena)

    def factor_top(self):
        sy = self.sym
        for s in self.top:
            c0, k, r, m = self._front_dims(s)
            F = self._arena[self.aoff[s]:self.aoff[s] + m * m].reshape(m, m).copy() if s in self.aoff else np.zeros((m, m))
            self._a_entries(s, F)                      # A is replicated input: every rank adds it itself
            for ch in self.children[s]:
                if sy["owner"][ch] < 0:
                    rl = self._rel(ch); F[np.ix_(rl, rl)] += self.cb[ch]
            self._eliminate(s, F)
        neg = sum(self.neg[s] for s in range(self.nsn) if sy["owner"][s] == self.rank)
        if self.rank == 0:
            neg += sum(self.neg[s] for s in self.top)
        return neg, 0

    def _fwd_front(self, s, extra):
        c0, k, r, m = self._front_dims(s)
        bs = np.zeros(m); bs[:k] = self.b[c0:c0 + k]
        bs += extra
        y = bs[:k]
        self.cvec[s] = bs[k:] - self.F21[s] @ (self.F11i[s] @ y)
        self.b[c0:c0 + k] = y

    def fwd_local(self, rhs):
        sy = self.sym
        self.b = rhs.numpy()[sy["perm"]].astype(float).copy()
        self.cvec = [None] * self.nsn
        for s in range(self.nsn):
            if sy["owner"][s] != self.rank:
                continue
            c0, k, r, m = self._front_dims(s)
            extra = np.zeros(m)
            for ch in self.children[s]:
                extra[self._rel(ch)] += self.cvec[ch]
            self._fwd_front(s, extra)
        self._toprhs[:] = 0.0
        for s in self.top:
            m = self._front_dims(s)[3]
            for ch in self.children[s]:
                if sy["owner"][ch] == self.rank:
                    self._toprhs[self.toff[s] + self._rel(ch)] += self.cvec[ch]

    def top_rhs(self):
        return torch.from_numpy(self._toprhs)

    def top_and_bwd(self, rhs):
        sy = self.sym
        for s in self.top:
            c0, k, r, m = self._front_dims(s)
            extra = self._toprhs[self.toff[s]:self.toff[s] + m].copy()
            for ch in self.children[s]:
                if sy["owner"][ch] < 0:
                    extra[self._rel(ch)] += self.cvec[ch]
            self._fwd_front(s, extra)
        x = np.zeros(self.n)
        mine = lambda s: sy["owner"][s] == self.rank or sy["owner"][s] < 0
        for s in range(self.nsn - 1, -1, -1):
            if not mine(s):
                continue
            c0, k, r, m = self._front_dims(s)
            x[c0:c0 + k] = self.F11i[s] @ (self.b[c0:c0 + k] - self.F21[s].T @ x[r[k:]])
        out = np.zeros(self.n)
        for s in range(self.nsn):
            if sy["owner"][s] == self.rank or (sy["owner"][s] < 0 and self.rank == 0):
                c0, k = sy["colptr"][s], sy["colptr"][s + 1] - sy["colptr"][s]
                out[sy["perm"][c0:c0 + k]] = x[c0:c0 + k]
        rhs.numpy()[:] = out

    def counters_tensor(self, neg, zero):
        return torch.tensor([neg, zero], dtype=torch.int64)

    def sync(self):
        pass

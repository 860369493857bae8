"""numpy stand-in for ipopt_amd.multigpu.HipEngine -- TEST SUPPORT ONLY (CPU, gloo).  Walks the symbolic structures
exported by the C ABI exactly as the HIP multi-GPU path does (numeric.hip: factor_dist / solve_dist): own subtrees first;
then, per exchange step (deepest ranges of ranks first), what the children OUTSIDE a replicated front's range contribute
is written into the front's arena square by ONE reporting rank per child, the driver all-reduces that step's part of the
arena, and the ranks of the range factor the front.  The classic replicated top is the case of one step; with the
subtree-to-subcube mapping (option subcube) a front is held only by the ranks beneath it.  Mirrored solve."""
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
        own, glo, gsz, gd, par = sy["owner"], sy["glo"], sy["gsz"], sy["gdepth"], sy["parent"]
        self.children = [[] for _ in range(self.nsn)]
        for s in range(self.nsn):
            if par[s] >= 0:
                self.children[par[s]].append(s)
        self.nsteps = int(gd[own < 0].max()) + 1 if (own < 0).any() else 1
        self.held = lambda s: own[s] < 0 and glo[s] <= self.rank < glo[s] + gsz[s]
        self.same = lambda a, b: own[a] < 0 and own[b] < 0 and glo[a] == glo[b] and gsz[a] == gsz[b]
        # a child reaches its replicated parent through the arena when the parent belongs to another range of ranks
        self.crosses = lambda c: par[c] >= 0 and own[par[c]] < 0 and not self.same(c, par[c])
        # ... and is reported by its owner / by the first rank of its range, after its own step (kind 0 = owned subtree root)
        self.reporter = lambda c: own[c] if own[c] >= 0 else glo[c]
        self.kind = lambda c: 0 if own[c] >= 0 else 1 + gd[c]
        self.step = [[s for s in range(self.nsn) if self.held(s) and gd[s] == d] for d in range(self.nsteps)]
        join = set(par[c] for c in range(self.nsn) if self.crosses(c))
        # layout: step by step and, inside a step, range of ranks by range of ranks -- the same on every rank (fronts of ranges this rank is
        # not in stay zero here); a square travels as its LOWER triangle packed by columns (numeric.hip: RangeSeg, arena_off)
        self.aoff, self.toff, self.abounds, self.tbounds, self.segs, a, t = {}, {}, [], [], [], 0, 0
        for d in range(self.nsteps):
            a0, t0 = a, t
            ranges = sorted(set((int(glo[s]), int(gsz[s])) for s in range(self.nsn) if own[s] < 0 and gd[s] == d))
            for lo, g in ranges:
                sa, st = a, t
                for s in range(self.nsn):
                    if own[s] < 0 and gd[s] == d and glo[s] == lo and gsz[s] == g:
                        m = int(sy["rowptr"][s + 1] - sy["rowptr"][s])
                        self.toff[s] = t; t += m
                        if s in join:
                            self.aoff[s] = a; a += m * (m + 1) // 2
                self.segs.append((d, lo, g, sa, a, st, t))
            self.abounds.append((a0, a)); self.tbounds.append((t0, t))
        self._arena = np.zeros(a); self._toprhs = np.zeros(t)

    def exchange_segments(self, d, what):
        """[(rank_lo, nranks_in_range, tensor)] of step d this rank takes part in (what 0: arena squares, 1: top right-hand sides)"""
        buf = self._arena if what == 0 else self._toprhs
        out = []
        for (sd, lo, g, a0, a1, t0, t1) in self.segs:
            if sd == d and lo <= self.rank < lo + g:
                b, e = (a0, a1) if what == 0 else (t0, t1)
                if e > b:
                    out.append((lo, g, torch.from_numpy(buf[b:e])))
        return out

    @staticmethod
    def _tril_index(m):
        # position of (i, c), i >= c, in the column-packed lower triangle
        c, i = np.triu_indices(m)            # pairs with c <= i, column by column
        return i, c

    def _square(self, s, m):
        """the (symmetric) m x m matrix an arena square stands for"""
        i, c = self._tril_index(m)
        F = np.zeros((m, m)); F[i, c] = self._arena[self.aoff[s]:self.aoff[s] + m * (m + 1) // 2]
        return F + np.tril(F, -1).T

    def num_steps(self):
        return self.nsteps

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

    def _report_arena(self, kind):
        sy = self.sym
        for ch in range(self.nsn):
            if self.crosses(ch) and self.kind(ch) == kind and self.reporter(ch) == self.rank:
                p = sy["parent"][ch]; m = self._front_dims(p)[3]
                F = np.zeros((m, m)); rl = self._rel(ch); F[np.ix_(rl, rl)] += self.cb[ch]
                i, c = self._tril_index(m)
                self._arena[self.aoff[p]:self.aoff[p] + m * (m + 1) // 2] += F[i, c]

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
        self._report_arena(0)

    def arena(self, d=0):
        a0, a1 = self.abounds[d]
        return torch.from_numpy(self._arena[a0:a1])

    def factor_step(self, d):
        for s in self.step[d]:
            c0, k, r, m = self._front_dims(s)
            F = self._square(s, m) if s in self.aoff else np.zeros((m, m))
            self._a_entries(s, F)                      # A is replicated input: every rank adds it itself
            for ch in self.children[s]:
                if self.same(ch, s):
                    rl = self._rel(ch); F[np.ix_(rl, rl)] += self.cb[ch]
            self._eliminate(s, F)
        if d > 0:
            self._report_arena(1 + d)

    def counters(self):
        sy = self.sym
        neg = sum(self.neg[s] for s in range(self.nsn) if sy["owner"][s] == self.rank or (sy["owner"][s] < 0 and sy["glo"][s] == self.rank))
        return neg, 0

    def _fwd_front(self, s, extra):
        c0, k, r, m = self._front_dims(s)
        bs = np.zeros(m); bs[:k] = self.b[c0:c0 + k]
        bs += extra
        y = bs[:k]
        self.cvec[s] = bs[k:] - self.F21[s] @ (self.F11i[s] @ y)
        self.b[c0:c0 + k] = y

    def _report_rhs(self, kind):
        sy = self.sym
        for ch in range(self.nsn):
            if self.crosses(ch) and self.kind(ch) == kind and self.reporter(ch) == self.rank:
                self._toprhs[self.toff[sy["parent"][ch]] + self._rel(ch)] += self.cvec[ch]

    def fwd_local(self, rhs):
        sy = self.sym
        self.b = rhs.numpy()[sy["perm"]].astype(float).copy()
        self.cvec = [None] * self.nsn
        self._toprhs[:] = 0.0
        for s in range(self.nsn):
            if sy["owner"][s] != self.rank:
                continue
            c0, k, r, m = self._front_dims(s)
            extra = np.zeros(m)
            for ch in self.children[s]:
                extra[self._rel(ch)] += self.cvec[ch]
            self._fwd_front(s, extra)
        self._report_rhs(0)

    def top_rhs(self, d=0):
        t0, t1 = self.tbounds[d]
        return torch.from_numpy(self._toprhs[t0:t1])

    def fwd_step(self, d):
        for s in self.step[d]:
            c0, k, r, m = self._front_dims(s)
            extra = self._toprhs[self.toff[s]:self.toff[s] + m].copy()
            for ch in self.children[s]:
                if self.same(ch, s):
                    extra[self._rel(ch)] += self.cvec[ch]
            self._fwd_front(s, extra)
        if d > 0:
            self._report_rhs(1 + d)

    def bwd(self, rhs):
        sy = self.sym
        x = np.zeros(self.n)
        mine = lambda s: sy["owner"][s] == self.rank or self.held(s)
        for s in range(self.nsn - 1, -1, -1):
            if not mine(s):
                continue
            c0, k, r, m = self._front_dims(s)
            x[c0:c0 + k] = self.F11i[s] @ (self.b[c0:c0 + k] - self.F21[s].T @ x[r[k:]])
        out = np.zeros(self.n)
        for s in range(self.nsn):
            if sy["owner"][s] == self.rank or (sy["owner"][s] < 0 and sy["glo"][s] == self.rank):
                c0, k = sy["colptr"][s], sy["colptr"][s + 1] - sy["colptr"][s]
                out[sy["perm"][c0:c0 + k]] = x[c0:c0 + k]
        rhs.numpy()[:] = out

    def counters_tensor(self, neg, zero):
        return torch.tensor([neg, zero], dtype=torch.int64)

    def sync(self):
        pass

"""Synthetic symmetric-indefinite KKT systems with inertia known BY CONSTRUCTION (SURVEY 8(c) pin 2,
8(d) item 4).  Pure numpy; used by the tests and by bench.py (no reference code involved).

All generators return (n, row, col, val, expected_neg) with 1-based triplets of ONE triangle,
possibly with duplicates and with entries in either triangle, exactly like the arrays Ipopt's
TripletHelper hands to a Triplet_Format backend (reference IpTripletHelper.cpp:805-842)."""
from __future__ import annotations

import numpy as np


class Xoshiro256:
    """xoshiro256** seeded through splitmix64 (tests/support/xoshiro256.c): ONE sequential stream of doubles uniform in [0, 1) -- the generator SURVEY.md 8(d)
    item 4 names for the synthetic system of BASELINE.json configs[3].  uniform(lo, hi, size) consumes prod(size) values, C order."""
    _lib = None

    def __init__(self, seed):
        import ctypes, os, subprocess
        if Xoshiro256._lib is None:
            here = os.path.dirname(os.path.abspath(__file__))
            so = os.path.join(here, "lib", "libxoshiro256.so")
            if not os.path.exists(so):
                subprocess.check_call(["make", "-s", "-C", here])
            Xoshiro256._lib = ctypes.CDLL(so)
            Xoshiro256._lib.xoshiro256_seed.argtypes = [ctypes.c_uint64, ctypes.c_void_p]
            Xoshiro256._lib.xoshiro256_fill.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        self.state = np.zeros(4, dtype=np.uint64)
        Xoshiro256._lib.xoshiro256_seed(int(seed), self.state.ctypes.data)

    def random(self, size):
        shape = (size,) if np.isscalar(size) else tuple(size)
        out = np.empty(int(np.prod(shape, dtype=np.int64)), dtype=np.float64)
        Xoshiro256._lib.xoshiro256_fill(self.state.ctypes.data, out.size, out.ctypes.data)
        return out.reshape(shape)

    def uniform(self, lo, hi, size):
        return lo + (hi - lo) * self.random(size)


def _finish(n, rows, cols, vals):
    return n, np.concatenate(rows).astype(np.int32) + 1, np.concatenate(cols).astype(np.int32) + 1, np.concatenate(vals).astype(np.float64)


def kkt_from_blocks(H_trip, Sigma, J_trip, dc, nx, m, h_upper=True):
    """K = [[H + diag(Sigma), J^T], [J, -diag(dc)]] as Ipopt lays it out: H entries (given upper when
    h_upper), a SEPARATE duplicate diagonal for Sigma, J strictly lower (row offset nx), explicit
    (2,2) diagonal."""
    hi, hj, hv = H_trip
    ji, jj, jv = J_trip
    rows = [np.minimum(hi, hj) if h_upper else np.maximum(hi, hj), np.arange(nx), ji + nx, np.arange(m) + nx]
    cols = [np.maximum(hi, hj) if h_upper else np.minimum(hi, hj), np.arange(nx), jj, np.arange(m) + nx]
    vals = [hv, Sigma, jv, -dc]
    return _finish(nx + m, rows, cols, vals)


def lukvl_like(n, seed=0, delta_c=0.0, sigma_scale=1.0):
    """Banded KKT with the sparsity of ScalableProblems LukVlE1 (reference
    examples/ScalableProblems/LuksanVlcek1.cpp:45-51,245-256): tridiagonal Hessian (upper), constraint
    i couples x_i, x_{i+1}, x_{i+2}; n variables, n-2 equality constraints.  H is made diagonally
    dominant so that the inertia is exactly (n, n-2, 0)."""
    rng = np.random.default_rng(seed)
    m = n - 2
    off = rng.uniform(-1, 1, n - 1)
    diag = np.zeros(n)
    diag[:-1] += np.abs(off); diag[1:] += np.abs(off)
    diag += rng.uniform(0.1, 1.0, n)
    hi = np.concatenate([np.arange(n), np.arange(n - 1)]); hj = np.concatenate([np.arange(n), np.arange(1, n)])
    hv = np.concatenate([diag, off])
    Sigma = sigma_scale * 10.0 ** rng.uniform(-3, 3, n)
    ji = np.repeat(np.arange(m), 3); jj = (np.arange(m)[:, None] + np.arange(3)[None, :]).ravel()
    jv = rng.uniform(-1, 1, (m, 3)); jv[:, 1] = 1.5 + rng.uniform(0, 1, m)   # anchor => full row rank
    return kkt_from_blocks((hi, hj, hv), Sigma, (ji, jj, jv.ravel()), np.full(m, delta_c), n, m) + (m,)


def grid_kkt(nx_grid, ny_grid, dof=1, ncon=1, seed=0, delta_c=0.0, sigma_exp=3.0, rng="pcg64"):
    """PDE-constrained-like KKT on an nx x ny grid (scaled down): `dof` primal
    unknowns and `ncon` (<= dof) constraints per node, 9-point coupling.  H SPD by diagonal
    dominance, J full row rank through a dominant anchor entry => inertia (n_x, m, 0) exactly.
    rng = "xoshiro": the values come from ONE xoshiro256** stream seeded by splitmix64(seed) (the bench workloads synth_1e6 /
    grid_1e5), consumed in this loop order: (1) the H blocks -- for the stencil offsets (di, dj) in (-1, 0, 1)^2 row-major, for the nodes p that have
    that neighbour q with q <= p, in the order i + nx j of a meshgrid walked i-major: dof x dof values of U(-1, 1), row-major; (2) one U(0.1, 1) per
    diagonal entry of H in the order the entries were generated; (3) Sigma: one U(-e, e) exponent per primal unknown; (4) J: for the stencil offsets and
    nodes in the same order (all neighbours, no q <= p filter), ncon x dof values 0.05 U(-1, 1); (5) an array of the same shape of 1 + U(0, 1), of which
    the anchor positions (p = q, constraint c on dof c) are used."""
    assert ncon <= dof
    rng = Xoshiro256(seed) if rng == "xoshiro" else np.random.default_rng(seed)
    N = nx_grid * ny_grid
    ii, jj_ = np.meshgrid(np.arange(nx_grid), np.arange(ny_grid), indexing="ij")
    pid = (ii + nx_grid * jj_).ravel()
    nbrs = []
    for di in (-1, 0, 1):
        for dj in (-1, 0, 1):
            a, b = ii + di, jj_ + dj
            ok = ((a >= 0) & (a < nx_grid) & (b >= 0) & (b < ny_grid)).ravel()
            nbrs.append((pid[ok], (a + nx_grid * b).ravel()[ok]))
    P = np.concatenate([p for p, _ in nbrs]); Q = np.concatenate([q for _, q in nbrs])
    keep = Q <= P
    P, Q = P[keep], Q[keep]
    # H blocks: dense dof x dof between neighbouring nodes (lower node pairs), symmetrised on the diagonal
    npair = P.shape[0]
    blk = rng.uniform(-1, 1, (npair, dof, dof))
    same = P == Q
    blk[same] = 0.5 * (blk[same] + np.transpose(blk[same], (0, 2, 1)))
    r = (P[:, None, None] * dof + np.arange(dof)[None, :, None]) + np.zeros((1, 1, dof), dtype=np.int64)
    c = (Q[:, None, None] * dof + np.arange(dof)[None, None, :]) + np.zeros((1, dof, 1), dtype=np.int64)
    r, c, v = r.ravel(), c.ravel(), blk.ravel()
    low = r >= c
    # for diagonal blocks keep only the lower triangle; for off-diagonal node pairs keep everything
    keep2 = low | np.repeat(~same, dof * dof)
    r, c, v = r[keep2], c[keep2], v[keep2]
    nxv = N * dof
    offd = r != c
    rowsum = np.zeros(nxv)
    np.add.at(rowsum, r[offd], np.abs(v[offd])); np.add.at(rowsum, c[offd], np.abs(v[offd]))
    dmask = ~offd
    v[dmask] = rowsum[r[dmask]] + rng.uniform(0.1, 1.0, dmask.sum())
    Sigma = 10.0 ** rng.uniform(-sigma_exp, sigma_exp, nxv)
    # J: row (p, cidx) touches all dofs of all 9 neighbours with small entries, anchor at (p, cidx)
    Pj = np.concatenate([p for p, _ in nbrs]); Qj = np.concatenate([q for _, q in nbrs])
    m = N * ncon
    jr = (Pj[:, None, None] * ncon + np.arange(ncon)[None, :, None]) + np.zeros((1, 1, dof), dtype=np.int64)
    jc = (Qj[:, None, None] * dof + np.arange(dof)[None, None, :]) + np.zeros((1, ncon, 1), dtype=np.int64)
    jv = 0.05 * rng.uniform(-1, 1, jr.shape)
    anchor = (Pj == Qj)[:, None, None] & (np.arange(ncon)[None, :, None] == np.arange(dof)[None, None, :])
    jv = np.where(anchor, 1.0 + rng.uniform(0, 1, jr.shape), jv)
    return kkt_from_blocks((r, c, v), Sigma, (jr.ravel(), jc.ravel(), jv.ravel()), np.full(m, delta_c), nxv, m) + (m,)


def to_scipy(n, row, col, val):
    """full symmetric scipy CSR matrix of a one-triangle triplet list (duplicates summed)."""
    import scipy.sparse as sp
    r, c = row - 1, col - 1
    lo, hi = np.minimum(r, c), np.maximum(r, c)
    L = sp.coo_matrix((val, (hi, lo)), shape=(n, n)).tocsr()
    D = sp.diags(L.diagonal())
    return (L + L.T - D).tocsr()


# ---- known-answer systems for the pivot-threshold contract (u = pivtol) ----
KAT_OPTS = dict(ordering=2, matching=0, scaling=0, nemin=1)     # natural order, no pre-pairing, no equilibration: the fronts are as written


def threshold_kat(eps=1e-5):
    """5 x 5 system whose first front has 3 fully-summed columns {1,2,3} and one update row {4} (natural order, KAT_OPTS).
    Column 1: tiny diagonal eps, coupling 1 to the UPDATE row, 1e-3 to column 2; column 2 couples strongly (1) to column 3.
    Bunch-Kaufman (which only sees the fully-summed block) prefers the 1x1 pivot eps.  With u = 1e-8 it passes the
    threshold test (eps >= u * 1), with u = 1e-4 it fails and the 2x2 pivot (1,2) -- which passes the MA57 test -- is taken
    instead: num_two differs, the inertia and the solution must not."""
    ent = [(1, 1, eps), (2, 1, 1e-3), (3, 1, 0.0), (4, 1, 1.0),
           (2, 2, 0.0), (3, 2, 1.0),
           (3, 3, 2.0), (4, 3, 0.5),
           (4, 4, -1.0), (5, 4, 1.0),
           (5, 5, 3.0)]
    r = np.array([e[0] for e in ent], dtype=np.int32); c = np.array([e[1] for e in ent], dtype=np.int32)
    return 5, r, c, np.array([e[2] for e in ent], dtype=np.float64)


def forced_pivot_kat(eps=1e-6):
    """4 x 4 system: the leaf front of variable 2 has ONE fully-summed column with diagonal eps and an update-row entry 1.  u = 1e-8 accepts the pivot; at u = 1e-4 it fails the threshold test, nothing else in the
    front can be pivoted on, and -- the structure being static -- it is eliminated anyway and reported: num_delay = 1."""
    ent = [(1, 1, 2.0), (3, 1, 1.0), (2, 2, eps), (4, 2, 1.0), (3, 3, -1.0), (4, 3, 0.5), (4, 4, -1.0)]
    r = np.array([e[0] for e in ent], dtype=np.int32); c = np.array([e[1] for e in ent], dtype=np.int32)
    return 4, r, c, np.array([e[2] for e in ent], dtype=np.float64)


def nearly_dependent_rows(delta=1e-15, nx=6):
    """KKT with two constraint rows that differ by `delta` (relative): numerically rank deficient for delta ~ eps.  delta_c = 0."""
    m = 3
    H = (np.arange(nx), np.arange(nx), np.full(nx, 2.0))
    ji = np.array([0, 0, 1, 1, 2, 2]); jj = np.array([0, 1, 0, 1, 3, 4]); jv = np.array([1.0, 2.0, 1.0, 2.0 * (1.0 + delta), 1.0, 1.0])
    return kkt_from_blocks(H, np.zeros(nx), (ji, jj, jv), np.zeros(m), nx, m)


def hostile_grid_kkt(nx_grid, ny_grid, dof=3, ncon=2, seed=0, frac=0.35, tiny=1e-6):
    """grid_kkt made hostile to static pivoting: a fraction `frac` of the Hessian diagonal entries (Sigma included) is replaced by
    +-tiny * U(0.1, 1) while the couplings stay O(1), delta_c = 0 -- the 1x1 candidates of those columns fail the threshold test at any
    practical u, 2x2 pivots are needed inside the pivot blocks, and in the separator fronts the rows below a pivot block see
    multipliers > 1/u (the a-posteriori check of the big fronts).  The inertia is whatever it is: the tests take it from the oracle
    (which delays) and from LAPACK on a dense copy."""
    n, r, c, v, m = grid_kkt(nx_grid, ny_grid, dof=dof, ncon=ncon, seed=seed, sigma_exp=0.0)
    rng = np.random.default_rng(seed + 1000)
    nxv = nx_grid * ny_grid * dof
    v = v.copy()
    diag = np.nonzero((r == c) & (r <= nxv))[0]            # (1-based triplets: Hessian diagonal)
    hit = diag[rng.random(diag.shape[0]) < frac]
    v[hit] = tiny * rng.uniform(0.1, 1.0, hit.shape[0]) * rng.choice([-1.0, 1.0], hit.shape[0])
    return n, r, c, v


def hostile_band_kkt(n, seed=0, frac=0.3, tiny=1e-6):
    """lukvl_like made hostile to static pivoting: no Sigma, a fraction `frac` of the Hessian diagonal at +-tiny * U(0.1, 1) against O(1)
    couplings.  All fronts have order <= ~20: the four-fronts-per-wavefront kernel (k_front_dpp16) must reject some of them and leave them to
    the strict kernel behind it.  Inertia from the oracle / LAPACK."""
    nn, r, c, v, m = lukvl_like(n, seed=seed, sigma_scale=0.0)
    rng = np.random.default_rng(seed + 2000)
    v = v.copy()
    diag = np.nonzero((r == c) & (r <= n))[0]
    hit = diag[rng.random(diag.shape[0]) < frac]
    v[hit] = tiny * rng.uniform(0.1, 1.0, hit.shape[0]) * rng.choice([-1.0, 1.0], hit.shape[0])
    return nn, r, c, v


def recorded_kkt(path, which=-1):
    """The KKT system of one factorisation of a boundary recording (tests/golden/*.kktrec, written by the reference's RecordingSolverInterface
    in oracle/ref_driver.cpp): (n, row, col, values, expected negative eigenvalues) as 1-based triplets of the lower triangle.  `which`
    counts the calls that carried a new matrix.  Reader kept apart from oracle/ (bench.py's GPU legs must not import the checker)."""
    buf = open(path, "rb").read()
    assert buf[:8] == b"KKTREC1\n"
    off, mats, ia, ja, fmt, dim = 8, [], None, None, 0, 0
    while off < len(buf):
        hdr = np.frombuffer(buf, dtype=np.int32, count=8, offset=off); off += 32
        if hdr[0] == 0:
            dim, nnz, fmt, nia = int(hdr[1]), int(hdr[2]), int(hdr[3]), int(hdr[4])
            ia = np.frombuffer(buf, dtype=np.int32, count=nia, offset=off).copy(); off += 4 * nia
            ja = np.frombuffer(buf, dtype=np.int32, count=nnz, offset=off).copy(); off += 4 * nnz
        else:
            dim, nnz, nrhs, newm, check, req, status = (int(v) for v in hdr[1:8])
            neg = int(np.frombuffer(buf, dtype=np.int32, count=1, offset=off)[0]); off += 4
            if newm:
                a = np.frombuffer(buf, dtype=np.float64, count=nnz, offset=off).copy(); off += 8 * nnz
                mats.append((a, neg, status))
            off += 16 * dim * nrhs
    a, neg, status = mats[which]
    if fmt == 0:                                   # triplets as recorded
        r, c = ia.astype(np.int32), ja.astype(np.int32)
    else:                                          # CSR of the upper triangle (EMatrixFormat 1 / 2: 0- / 1-offset): row = position in ia
        base = 0 if fmt == 1 else 1
        rows = np.repeat(np.arange(dim, dtype=np.int32), np.diff(ia)) + 1
        r, c = rows, (ja - base + 1).astype(np.int32)
    return dim, r, c, a, neg

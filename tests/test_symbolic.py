"""-m 'not gpu': host logic.  The symbolic analysis (canonicalisation of duplicate / mixed-triangle
triplets, 2x2 pre-pairing, ordering, supernodes, relative indices, scatter map, level schedule,
multi-GPU ownership) is validated by walking its exported structures with a numpy block-multifrontal
(tests/support/mirror.py) and comparing with scipy / by-construction inertia."""
import numpy as np
import pytest

import ipopt_amd
from tests.support import kktgen, mirror

CASES = {
    "lukvl300": lambda dc=0.0: kktgen.lukvl_like(300, seed=1, delta_c=dc),
    "lukvl300_dc": lambda dc=1e-8: kktgen.lukvl_like(300, seed=2, delta_c=dc),
    "grid9x7": lambda dc=0.0: kktgen.grid_kkt(9, 7, dof=2, ncon=1, seed=3, delta_c=dc),
    "grid16x16_d3c2": lambda dc=0.0: kktgen.grid_kkt(16, 16, dof=3, ncon=2, seed=4, delta_c=dc),
}
OPTS = [dict(), dict(ordering=1), dict(ordering=2, nemin=1), dict(matching=0), dict(nemin=32, max_sn_cols=128), dict(nd_leaf=16, nemin=2)]


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("opts", OPTS, ids=[str(o) for o in OPTS])
def test_symbolic_structures_solve_the_system(case, opts):
    # without the 2x2 pre-pairing a zero (2,2) diagonal is a zero pivot by construction: exercise that
    # option on the quasi-definite variant (delta_c = 1) -- this test is about structures, not pivoting
    n, r, c, v, neg = CASES[case](1.0) if not opts.get("matching", 1) else CASES[case]()
    s = ipopt_amd.KKTSolver(**opts)
    s.initialize_structure(n, r, c, vals=v)
    sym = mirror.fetch(s)
    I = sym["info"]
    assert sorted(sym["perm"].tolist()) == list(range(n))
    assert I.nnz_l >= I.nnz_a and I.num_levels >= 1 and I.maxfront <= n
    # level schedule: every child strictly below its parent; buckets cover every supernode once
    par, lev = sym["parent"], sym["level"]
    for sn in range(I.num_sn):
        if par[sn] >= 0:
            assert lev[sn] < lev[par[sn]]
    lsn = s.symbolic(14, I.num_sn)
    assert sorted(lsn.tolist()) == list(range(I.num_sn))
    K = kktgen.to_scipy(n, r, c, v)
    b = K @ np.linspace(1.0, 2.0, n)
    x, nneg = mirror.factor_solve(sym, v, b)
    assert nneg == neg
    assert np.abs(K @ x - b).max() <= 1e-9 * np.abs(b).max()
    # pre-pairing: pairs are mutual, adjacent in the elimination order and inside one supernode
    pair = sym["pair"]
    iperm = np.empty(n, dtype=np.int64); iperm[sym["perm"]] = np.arange(n)
    snof = np.repeat(np.arange(I.num_sn), np.diff(sym["colptr"]))
    for i in np.where(pair >= 0)[0]:
        j = pair[i]
        assert pair[j] == i and abs(iperm[i] - iperm[j]) == 1 and snof[iperm[i]] == snof[iperm[j]]
    if opts.get("matching", 1):
        assert I.num_pairs == neg    # every zero-diagonal constraint row found a partner


def test_duplicates_and_mixed_triangles_are_summed():
    """SURVEY 8(b) pitfall 1: Ipopt's triplets always contain duplicates (diag(W) and D_x) and may list an
    entry in either triangle; both must be canonicalised (reference IpTripletToCSRConverter.cpp:352-359)."""
    n, r, c, v, _ = kktgen.lukvl_like(50, seed=5)
    rng = np.random.default_rng(0)
    # split every value into two duplicates, flip triangles at random, shuffle
    r2 = np.concatenate([r, r]); c2 = np.concatenate([c, c]); v2 = np.concatenate([0.25 * v, 0.75 * v])
    flip = rng.random(r2.shape[0]) < 0.5
    r2[flip], c2[flip] = c2[flip].copy(), r2[flip].copy()
    p = rng.permutation(r2.shape[0]); r2, c2, v2 = r2[p], c2[p], v2[p]
    s1 = ipopt_amd.KKTSolver(); s1.initialize_structure(n, r, c, vals=v)
    s2 = ipopt_amd.KKTSolver(); s2.initialize_structure(n, r2, c2, vals=v2)
    assert s1.info().nnz_a == s2.info().nnz_a
    K = kktgen.to_scipy(n, r, c, v); b = K @ np.ones(n)
    x2, _ = mirror.factor_solve(mirror.fetch(s2), v2, b)
    assert np.abs(x2 - 1).max() < 1e-8


@pytest.mark.parametrize("case", ["grid_dups", "band", "csr"])
def test_duplicate_lists_hold_every_triplet_of_a_slot_in_ascending_order(case):
    """k_gather_values sums the triplets of a slot in the order of these lists (bitwise reproducible values): every triplet appears exactly once,
    under the slot the triplet map names, in ascending triplet index; slots nobody supplied (an absent diagonal) have an empty list."""
    if case == "grid_dups":
        n, r, c, v, _ = kktgen.grid_kkt(14, 11, dof=2, ncon=1, seed=3)
        rng = np.random.default_rng(0)
        extra = rng.integers(0, len(v), 300)                                  # duplicates, some of them with the triangle swapped
        r2, c2 = r[extra].copy(), c[extra].copy()
        sw = rng.random(300) < 0.5
        r2[sw], c2[sw] = c[extra][sw], r[extra][sw]
        r, c, v = np.concatenate([r, r2]), np.concatenate([c, c2]), np.concatenate([v, rng.standard_normal(300)])
        perm = rng.permutation(len(v)); r, c, v = r[perm].copy(), c[perm].copy(), v[perm].copy()
    elif case == "band":
        n, r, c, v, _ = kktgen.lukvl_like(700, seed=4)
    else:
        n, r, c, v, _ = kktgen.grid_kkt(9, 7, dof=2, ncon=1, seed=5)
    s = ipopt_amd.KKTSolver()
    if case == "csr":
        K = kktgen.to_scipy(n, r, c, v)
        import scipy.sparse as sp
        U = sp.triu(K, format="csr"); U.sort_indices()
        s.initialize_structure(n, (U.indptr + 1).astype(np.int32), (U.indices + 1).astype(np.int32), fmt=1)
        nnz = U.nnz
    else:
        s.initialize_structure(n, r, c, vals=v)
        nnz = len(v)
    I = s.info()
    t2s, dptr, dsrc = s.symbolic(9, nnz), s.symbolic(21, I.nnz_a + 1), s.symbolic(22, nnz)
    assert dptr[0] == 0 and dptr[-1] == nnz and np.all(np.diff(dptr) >= 0)
    assert sorted(dsrc.tolist()) == list(range(nnz))
    slot_of = np.repeat(np.arange(I.nnz_a), np.diff(dptr))
    assert np.array_equal(t2s[dsrc], slot_of)
    inside = np.diff(dsrc) > 0
    same = slot_of[1:] == slot_of[:-1]
    assert np.all(inside[same])
    if case == "grid_dups":
        assert np.diff(dptr).max() >= 2


def test_csr_upper_format_equals_triplet():
    n, r, c, v, _ = kktgen.grid_kkt(6, 5, dof=1, ncon=1, seed=6)
    K = kktgen.to_scipy(n, r, c, v)
    import scipy.sparse as sp
    U = sp.triu(K).tocsr(); U.sort_indices()
    s = ipopt_amd.KKTSolver()
    s.initialize_structure(n, (U.indptr + 1).astype(np.int32), (U.indices + 1).astype(np.int32), fmt=1, vals=U.data)
    b = K @ np.ones(n)
    x, _ = mirror.factor_solve(mirror.fetch(s), U.data, b)
    assert np.abs(x - 1).max() < 1e-8


def test_edge_cases_empty_diagonal_and_ragged():
    s = ipopt_amd.KKTSolver()
    s.initialize_structure(0, np.zeros(0, np.int32), np.zeros(0, np.int32))
    assert s.info().n == 0 and s.info().num_sn == 0
    # purely diagonal matrix given with a missing diagonal entry (structurally singular row is allowed)
    s = ipopt_amd.KKTSolver()
    s.initialize_structure(4, np.array([1, 2, 4], np.int32), np.array([1, 2, 4], np.int32), vals=np.array([1.0, -2.0, 3.0]))
    I = s.info()
    assert I.n == 4 and I.nnz_a == 4 and I.nnz_l == 4
    # one dense row (arrow matrix): everything ends up in few fronts, still valid
    n = 40
    r = np.concatenate([np.arange(1, n + 1), np.full(n - 1, n)]).astype(np.int32)
    c = np.concatenate([np.arange(1, n + 1), np.arange(1, n)]).astype(np.int32)
    v = np.concatenate([np.full(n, 4.0), np.full(n - 1, 0.1)])
    s = ipopt_amd.KKTSolver(); s.initialize_structure(n, r, c, vals=v)
    K = kktgen.to_scipy(n, r, c, v); b = K @ np.ones(n)
    x, neg = mirror.factor_solve(mirror.fetch(s), v, b)
    assert neg == 0 and np.abs(x - 1).max() < 1e-10
    with pytest.raises(ipopt_amd.KKTError):
        ipopt_amd.KKTSolver().initialize_structure(3, np.array([1, 5], np.int32), np.array([1, 1], np.int32))


@pytest.mark.parametrize("nranks", [2, 4, 8])
def test_multigpu_ownership_is_a_subtree_partition(nranks):
    n, r, c, v, _ = kktgen.grid_kkt(24, 24, dof=2, ncon=1, seed=7)
    s = ipopt_amd.KKTSolver(nranks=nranks)
    s.initialize_structure(n, r, c, vals=v)
    sym = mirror.fetch(s)
    own, par = sym["owner"], sym["parent"]
    assert set(np.unique(own)) <= set(range(-1, nranks)) and (own >= 0).any()
    for sn in range(sym["info"].num_sn):
        p = par[sn]
        if own[sn] == -1:
            assert p < 0 or own[p] == -1            # the replicated top is closed upwards
        elif p >= 0:
            assert own[p] in (own[sn], -1)          # a subtree never crosses ranks
    assert len(set(own[own >= 0].tolist())) == nranks


@pytest.mark.parametrize("nranks", [2, 3, 4, 6, 8])
def test_subtree_to_subcube_mapping_is_a_nested_partition(nranks):
    """option subcube: a replicated front is held by a RANGE of ranks [glo, glo + gsz): the ranks beneath it.  Ranges are nested along the
    tree (a child's range lies inside its parent's), a range is cut deeper (gdepth) exactly when it shrinks, owned subtrees hang below
    the range that contains their rank, in-place chains stay inside one range -- and the most loaded rank carries less than with one top
    replicated on every rank."""
    n, r, c, v, _ = kktgen.grid_kkt(40, 36, dof=2, ncon=1, seed=7)

    def load(sym, P):
        k = np.diff(sym["colptr"]).astype(float); m = np.diff(sym["rowptr"]).astype(float)
        w = k * m * m
        return max(w[(sym["glo"] <= rk) & (rk < sym["glo"] + sym["gsz"])].sum() for rk in range(P)), w.sum()

    syms = {}
    for sub in (0, 1):
        s = ipopt_amd.KKTSolver(nranks=nranks, subcube=sub)
        s.initialize_structure(n, r, c, vals=v)
        syms[sub] = mirror.fetch(s)
        syms[sub]["alias"] = s.symbolic(17, syms[sub]["info"].num_sn)
    sym = syms[1]
    own, par, glo, gsz, gd, alias = sym["owner"], sym["parent"], sym["glo"], sym["gsz"], sym["gdepth"], sym["alias"]
    assert len(set(own[own >= 0].tolist())) == nranks and (own < 0).any()
    for sn in range(sym["info"].num_sn):
        p = par[sn]
        assert 0 <= glo[sn] and glo[sn] + gsz[sn] <= nranks and gsz[sn] >= 1
        if own[sn] >= 0:
            assert (glo[sn], gsz[sn]) == (own[sn], 1)
            assert p < 0 or own[p] == own[sn] or (own[p] < 0 and glo[p] <= own[sn] < glo[p] + gsz[p])
        else:
            assert gsz[sn] >= 2
            if p >= 0:
                assert own[p] < 0 and glo[p] <= glo[sn] and glo[sn] + gsz[sn] <= glo[p] + gsz[p]
                same = (glo[p], gsz[p]) == (glo[sn], gsz[sn])
                assert gd[p] == gd[sn] if same else gd[p] < gd[sn]
            else:
                assert (glo[sn], gsz[sn], gd[sn]) == (0, nranks, 0)
        if alias[sn] >= 0:
            a = alias[sn]
            assert own[a] == own[sn] and (glo[a], gsz[a]) == (glo[sn], gsz[sn])
    # ranges of one depth are disjoint or equal
    rng = sorted(set((int(gd[sn]), int(glo[sn]), int(gsz[sn])) for sn in range(sym["info"].num_sn) if own[sn] < 0))
    for d, a, g in rng:
        for d2, a2, g2 in rng:
            if d2 == d and (a2, g2) != (a, g):
                assert a + g <= a2 or a2 + g2 <= a
    # the classic mapping reports every replicated front on every rank
    s0 = syms[0]
    assert ((s0["glo"][s0["owner"] < 0] == 0) & (s0["gsz"][s0["owner"] < 0] == nranks) & (s0["gdepth"][s0["owner"] < 0] == 0)).all()
    if nranks >= 4:
        (l1, tot), (l0, _) = load(sym, nranks), load(s0, nranks)
        assert l1 < l0 and tot / l1 > 0.4 * nranks


def test_concurrent_analyses_share_the_recycling_allocator():
    """several handles analysed at the same time from different threads (ctypes drops the GIL): the block cache of the analysis is shared,
    reference-counted by the running analyses and must neither mix up blocks nor lose them -- same permutations as one after the other, and
    the resident memory of the process returns to where it was once the handles are gone"""
    import gc
    import threading

    def rss():
        return int(open("/proc/self/statm").read().split()[1]) * 4096

    cases = [kktgen.grid_kkt(70 + 9 * i, 60 + 5 * i, dof=3, ncon=2, seed=i) for i in range(4)]      # big enough for arrays beyond the cache threshold
    serial = []
    for n, r, c, v, _ in cases:
        s = ipopt_amd.KKTSolver(); s.initialize_structure(n, r, c, vals=v); serial.append(s.symbolic(0, n).copy()); del s
    gc.collect(); base = rss()
    out, keep = [None] * 4, [None] * 4

    def work(i):
        n, r, c, v, _ = cases[i]
        for _ in range(3):
            s = ipopt_amd.KKTSolver(); s.initialize_structure(n, r, c, vals=v)
            out[i] = s.symbolic(0, n).copy(); keep[i] = s

    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in th]; [t.join() for t in th]
    for i in range(4):
        assert np.array_equal(out[i], serial[i])
    keep[:] = [None] * 4
    gc.collect()
    assert rss() <= base + (64 << 20)


def test_chain_groups_are_consistent():
    """in-place chains and chain groups (numeric.hip relies on these invariants): an in-place front has exactly its chain
    child's update rows; a group is <= 4 consecutive links on consecutive levels with <= 256 columns; grp_rem = columns
    of the later links of the group"""
    n, r, c, v, neg = kktgen.grid_kkt(40, 36, dof=3, ncon=2, seed=5)
    s = ipopt_amd.KKTSolver(device=-1)
    s.initialize_structure(n, r, c, vals=v)
    I = s.info(); nsn = I.num_sn
    cp, rp = s.symbolic(1, nsn + 1), s.symbolic(2, nsn + 1)
    rows, par, lev = s.symbolic(3, rp[-1]), s.symbolic(4, nsn), s.symbolic(5, nsn)
    gpos, grem, alias, cls = s.symbolic(15, nsn), s.symbolic(16, nsn), s.symbolic(17, nsn), s.symbolic(23, nsn)
    k, m = np.diff(cp), np.diff(rp)
    assert (alias >= 0).sum() > 0 and gpos.max() >= 1, "the test matrix must produce separator chains"
    assert np.array_equal(cls, np.where(m <= 32, 0, np.where(m <= 64, 1, np.where(m <= 128, 2, 3))))      # kernel class of a front = its order's
    nxt = {}
    for p in range(nsn):
        ch = alias[p]
        if ch < 0:
            assert gpos[p] == 0
            continue
        assert par[ch] == p and cls[p] == 3 and cls[ch] == 3
        assert np.array_equal(rows[rp[ch] + k[ch]:rp[ch + 1]], rows[rp[p]:rp[p + 1]])       # the front IS the child's update block
        if gpos[p] > 0:
            assert gpos[p] == gpos[ch] + 1 and lev[p] == lev[ch] + 1
            nxt[ch] = p
    for sn in range(nsn):
        if gpos[sn] == 0:                    # walk the group from its head
            cols, cur, links = k[sn], sn, 1
            while cur in nxt:
                assert grem[cur] == grem[nxt[cur]] + k[nxt[cur]]
                cur = nxt[cur]; cols += k[cur]; links += 1
            assert grem[cur] == 0 and links <= 4 and cols <= 256


@pytest.mark.parametrize("nx,ny,limit", [(80, 50, 510), (64, 64, 540), (120, 40, 600)])
def test_nested_dissection_finds_straight_separators_on_stencil_kkt(nx, ny, limit):
    """ordering quality guard: on a 9-point-coupled grid KKT (5 unknowns per node) the separators must be straight grid lines
    (level structures rooted at the last level, or at half of it on square domains, of a pseudo-peripheral BFS).  Largest
    front with straight cuts: 460 / 490 / 545 rows; the L-shaped level sets of the plain BFS give 700-800."""
    n, r, c, v, neg = kktgen.grid_kkt(nx, ny, dof=3, ncon=2, seed=7)
    s = ipopt_amd.KKTSolver(device=-1)
    s.initialize_structure(n, r, c, vals=v)
    assert s.info().maxfront <= limit


@pytest.mark.parametrize("shape", [(60, 44), (160, 130)], ids=["serial-dissection", "task-pool"])
def test_analysis_is_independent_of_the_thread_count(monkeypatch, shape):
    """the task-parallel nested dissection must give the same permutation whatever the number of host threads and however the pool
    happens to schedule the pieces (Ipopt runs must be reproducible from one machine to the next).  The larger grid has more than
    50 000 compressed nodes: there the bisections run on the shared task pool from the second piece on."""
    n, r, c, v, neg = kktgen.grid_kkt(shape[0], shape[1], dof=3, ncon=2, seed=9)
    perms = []
    for th in ("1", "3", "8", "8"):
        monkeypatch.setenv("MI355X_KKT_THREADS", th)
        s = ipopt_amd.KKTSolver(device=-1)
        s.initialize_structure(n, r, c, vals=v)
        perms.append(s.symbolic(0, n).copy())
    assert all(np.array_equal(perms[0], q) for q in perms[1:])


def test_side_children_of_chain_links_hang_down_the_chain(monkeypatch, golden_dir):
    """finish_analysis 9b (DESIGN "Round 5, second half"): on the MBndryCntrl_3D family two of three links of the in-place separator chains have a dangling side
    child of a few rows; the assembly tree hangs it on a lower link of the chain (the links above come out pure: chain groups, rank-256 updates).  The
    elimination order, the row structures and the levels must not change, every child's update rows must lie in its (new) parent's front, and the
    numpy block-multifrontal that walks the exported structures must still factor and solve the system -- here the first KKT matrix of the reference's own
    MBndryCntrl_3D N = 14 run (tests/golden/mbndry3d_14.kktrec, tests/golden/make_golden.sh)."""
    import os
    n, r, c, v, neg = kktgen.recorded_kkt(os.path.join(golden_dir, "mbndry3d_14.kktrec"), which=0)
    syms = {}
    for off in (False, True):
        if off: monkeypatch.setenv("MI355X_KKT_DISABLE", "purify")
        else: monkeypatch.delenv("MI355X_KKT_DISABLE", raising=False)
        s = ipopt_amd.KKTSolver()
        s.initialize_structure(n, r, c, vals=v)
        syms[off] = mirror.fetch(s)
    a, b = syms[False], syms[True]
    for key in ("perm", "colptr", "rowptr", "rows", "level"):
        assert np.array_equal(a[key], b[key]), key
    moved = np.where(a["parent"] != b["parent"])[0]
    assert len(moved) >= 20                                   # (51 at the time of writing)
    colptr, rowptr, rows, par, lev = a["colptr"], a["rowptr"], a["rows"], a["parent"], a["level"]
    m = np.diff(rowptr); k = np.diff(colptr)
    for ch in moved:
        p_new, p_nat = par[ch], b["parent"][ch]
        assert lev[ch] < lev[p_new] <= lev[p_nat] and m[p_new] > 128          # a lower link of a chain of BIG fronts, still above the child
        upd = rows[rowptr[ch] + k[ch]:rowptr[ch + 1]]
        prow = rows[rowptr[p_new] + k[p_new]:rowptr[p_new + 1]]
        assert np.isin(upd, prow).all()                          # the child's rows lie in the UPDATE part of its new parent: it rides up inside the chain's blocks
    # the natural parents of the moved children have come out pure where they continue a chain: fewer links with more than one child
    nch = lambda sym: np.bincount(sym["parent"][sym["parent"] >= 0], minlength=len(m))
    big = m > 128
    assert (nch(a)[big] > 1).sum() < (nch(b)[big] > 1).sum()
    K = kktgen.to_scipy(n, r, c, v)
    rhs = K @ np.linspace(1.0, 2.0, n)
    x, nneg = mirror.factor_solve(a, v, rhs)
    assert nneg == neg
    assert np.abs(K @ x - rhs).max() <= 1e-9 * np.abs(rhs).max()


def test_delays_out_of_rehung_children_follow_the_elimination_tree(golden_dir):
    """A failed pivot of a side child that 9b has hung on a lower chain link must still be delayed to the front that ELIMINATES later than the child (its parent in
    the elimination tree), not to the assembly parent: the edited structure is walked by the numpy multifrontal, which must factor and solve the system."""
    import os
    n, r, c, v, neg = kktgen.recorded_kkt(os.path.join(golden_dir, "mbndry3d_14.kktrec"), which=0)
    s = ipopt_amd.KKTSolver()
    s.initialize_structure(n, r, c, vals=v)
    a = mirror.fetch(s)
    colptr, rowptr, rows, par = a["colptr"], a["rowptr"], a["rows"], a["parent"]
    k = np.diff(colptr)
    # rehung children: the parent does not hold the child's first update row among its own columns
    first_upd = np.array([rows[rowptr[i] + k[i]] if rowptr[i] + k[i] < rowptr[i + 1] else -1 for i in range(len(k))])
    snof = np.repeat(np.arange(len(k)), k)
    rehung = [i for i in range(len(k)) if par[i] >= 0 and first_upd[i] >= 0 and snof[first_upd[i]] != par[i]]
    assert len(rehung) >= 20
    cols = np.array([a["perm"][colptr[i]] for i in rehung[:12]], dtype=np.int32) + 1        # one column of each, caller's (1-based) numbering
    assert s.delay_columns(cols) == len(cols)
    e = mirror.fetch(s)
    assert sorted(e["perm"].tolist()) == list(range(n))
    for sn in range(e["info"].num_sn):
        if e["parent"][sn] >= 0:
            assert e["level"][sn] < e["level"][e["parent"][sn]]
    # every delayed column now sits in the supernode that held its child's first update row (or a piece of it, if that supernode was cut at 64 columns)
    iperm = np.empty(n, dtype=np.int64); iperm[e["perm"]] = np.arange(n)
    K = kktgen.to_scipy(n, r, c, v)
    rhs = K @ np.linspace(1.0, 2.0, n)
    x, nneg = mirror.factor_solve(e, v, rhs)
    assert nneg == neg
    assert np.abs(K @ x - rhs).max() <= 1e-9 * np.abs(rhs).max()


def test_row_view_lists_every_entry_of_both_triangles_by_ascending_other_index():
    """The symmetric row view (selectors 24-26): row i holds every entry (i, c) and (r, i) of the permuted lower pattern once, by ascending other index,
    with the CSC slot it came from -- what the equilibration sweeps and the device refinement gather over."""
    n, r, c, v, _ = kktgen.grid_kkt(9, 7, dof=2, ncon=1, seed=3)
    s = ipopt_amd.KKTSolver()
    s.initialize_structure(n, r, c, vals=v)
    acolptr = s.symbolic(7, n + 1); arow = s.symbolic(8, int(acolptr[-1]))
    acol = np.repeat(np.arange(n), np.diff(acolptr))
    ptr = s.symbolic(24, n + 1); L = int(ptr[-1]); idx = s.symbolic(25, L); col = s.symbolic(26, L)
    nd = int((arow != acol).sum())
    assert L == len(arow) + nd
    for i in range(n):
        sl, oc = idx[ptr[i]:ptr[i + 1]], col[ptr[i]:ptr[i + 1]]
        assert (np.diff(oc) > 0).all()                                         # ascending other index, each once
        for q, o in zip(sl, oc):
            assert {int(arow[q]), int(acol[q])} == {i, int(o)}


@pytest.mark.parametrize("source", ["mbndry3d_14", "grid40x36"])
def test_recycled_contribution_blocks_never_share_space_while_both_are_alive(source, monkeypatch, golden_dir):
    """Step 12a of the analysis (round 6): blocks that only carry a contribution to their parent are laid out over the LEVEL schedule -- a block is
    written at its front's level, read at its assembly parent's level, and its space may be written again `window` + 1 levels later (the numeric
    schedule joins its look-ahead streams every `window` levels).  Checked on the exported plan: (1) with the plan off every block has its own space;
    (2) with it on, two blocks whose address ranges intersect have disjoint [born, consumed + window] level intervals, blocks that host an in-place
    chain (factor storage) and blocks of small fronts intersect nothing; (3) the permutation, supernodes and row structures are those of the plain layout."""
    import os
    if source == "mbndry3d_14":
        n, r, c, v, neg = kktgen.recorded_kkt(os.path.join(golden_dir, "mbndry3d_14.kktrec"), which=0)
    else:
        n, r, c, v, neg = kktgen.grid_kkt(40, 36, dof=2, ncon=1, seed=8)

    def plan(mode):
        monkeypatch.setenv("MI355X_KKT_RECYCLE", mode)
        s = ipopt_amd.KKTSolver(device=-1)
        s.initialize_structure(n, r, c, vals=v)
        I = s.info(); N = I.num_sn
        p = s.symbolic(27, 5).astype(np.int64)
        o = s.symbolic(28, 2 * N).astype(np.int64)
        off = (o[0::2] & 0xffffffff) | (o[1::2] << 32)
        return s, I, int(p[0]), ((p[1] & 0xffffffff) | (p[2] << 32), (p[3] & 0xffffffff) | (p[4] << 32)), off

    s0, I0, w0, (plain0, res0), off0 = plan("0")
    s1, I1, w1, (plain1, res1), off1 = plan("1")
    N = I0.num_sn
    assert w0 == 0 and w1 > 0 and I0.cb_doubles == plain0 == plain1 and I1.cb_doubles <= I0.cb_doubles and res1 <= I1.cb_doubles
    for sel, cnt in ((0, n), (1, N + 1), (2, N + 1), (4, N), (5, N), (17, N)):
        assert np.array_equal(s0.symbolic(sel, cnt), s1.symbolic(sel, cnt))
    colptr, rowptr = s1.symbolic(1, N + 1).astype(np.int64), s1.symbolic(2, N + 1).astype(np.int64)
    parent, lev, alias, cls = s1.symbolic(4, N), s1.symbolic(5, N), s1.symbolic(17, N), s1.symbolic(23, N)
    mu = np.diff(rowptr) - np.diff(colptr)
    own = np.where((alias < 0) & (mu > 0))[0]
    host = np.zeros(N, bool); host[alias[alias >= 0]] = True
    born = lev[own]
    dead = np.where(parent[own] >= 0, lev[np.maximum(parent[own], 0)], lev.max() + 1) + w1      # last level at which the space still belongs to the block
    lo, hi = off1[own], off1[own] + mu[own].astype(np.int64) ** 2
    order = np.argsort(lo, kind="stable")
    shared = 0
    for a_i, a in enumerate(order):
        for b in order[a_i + 1:]:
            if lo[b] >= hi[a]:
                break
            # the two ranges intersect: they may not be alive together, and neither may be a chain host or a small front's block
            shared += 1
            assert born[b] > dead[a] or born[a] > dead[b], (own[a], own[b])
            assert not host[own[a]] and not host[own[b]] and cls[own[a]] == 3 and cls[own[b]] == 3
    if source == "mbndry3d_14":
        assert shared > 0 and I1.cb_doubles < I0.cb_doubles      # (3-D: the plan does reuse space; the 2-D grid keeps everything within the window)
    # plain layout: nothing intersects
    lo0 = off0[own]; o0 = np.argsort(lo0, kind="stable")
    assert np.all(lo0[o0][1:] >= (lo0 + mu[own].astype(np.int64) ** 2)[o0][:-1])


def test_a_delayed_column_waits_up_a_full_chain_link_instead_of_cutting_it():
    """restructure_delays, round 6: a column delayed into a FULL (max_sn_cols = 64) link of a separator chain used to cut that link into 64 + 1 -- a one-column
    link with launches of its own, and one more tree level for every ancestor, out of step with its siblings (one edit of 100 columns: 53 -> 57 levels and a
    36 % slower factorisation at KKT dimension 2 * 10^5).  Now it waits further up the chain until a link has room: same number of supernodes, same number of
    levels, no supernode above 64 columns, a valid permutation, and the column sits in a supernode ABOVE the one it left."""
    n, r, c, v, neg = kktgen.grid_kkt(120, 100, dof=3, ncon=2, seed=77, sigma_exp=6.0)
    s = ipopt_amd.KKTSolver(device=-1, delay_rounds=0)
    s.initialize_structure(n, r, c, vals=v)
    I0 = s.info(); N0 = I0.num_sn
    colptr, rowptr, parent, perm = s.symbolic(1, N0 + 1), s.symbolic(2, N0 + 1), s.symbolic(4, N0), s.symbolic(0, n)
    k = np.diff(colptr); m = np.diff(rowptr)
    # links whose parent is the next supernode, FULL (64 columns) and a front of 256 rows and more: a column delayed out of such a link used to cut its parent
    cand = [sn for sn in range(N0 - 1) if k[sn + 1] == 64 and m[sn + 1] >= 256 and parent[sn] == sn + 1]
    assert len(cand) >= 8
    rng = np.random.default_rng(3)
    picks = rng.choice(cand, size=8, replace=False)
    cols = np.array([perm[colptr[sn] + int(rng.integers(0, k[sn]))] for sn in picks]) + 1          # caller's numbering, 1-based
    moved = s.delay_columns(cols)
    I1 = s.info(); N1 = I1.num_sn
    assert moved == 8 and N1 == N0 and I1.num_levels == I0.num_levels
    colptr1, perm1 = s.symbolic(1, N1 + 1), s.symbolic(0, n)
    assert np.diff(colptr1).max() <= 64 and sorted(perm1.tolist()) == list(range(n))
    iperm1 = np.empty(n, dtype=np.int64); iperm1[perm1] = np.arange(n)
    for sn, col in zip(picks, cols - 1):
        sn_new = int(np.searchsorted(colptr1, iperm1[col], side="right") - 1)
        assert sn_new > sn + 1                       # (the parent link was full: the column went further up)
    # an edit that moves MANY columns (more than n / 64) keeps the old rule: fronts may be cut
    many = rng.choice(n, size=n // 32, replace=False) + 1
    s.delay_columns(many)
    assert np.diff(s.symbolic(1, s.info().num_sn + 1)).max() <= 64

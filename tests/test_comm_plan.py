"""First contact with N GPUs, without having them (CPU): the collectives the C library will issue on every rank.

The distributed factorisation / solve (numeric.hip factor_dist / solve_dist) and the sub-communicator set-up of mi355x_kkt_set_comm_rccl
(make_subcomms: one ncclCommSplit per exchange step that has a range smaller than the machine) are driven by the symbolic structure alone.
mi355x_kkt_comm_plan walks that SAME code path on the host -- no device -- and lists, for one rank, every ncclCommSplit and every all-reduce in
issue order.  What makes a multi-rank run deadlock-free is that the ranks of a communicator issue the same collectives, with the same counts,
in an order that can be matched: this test plays all ranks' lists against each other for worlds of 2..8 ranks, both mappings, both the
range-local schedule and its whole-communicator fall-back.  (Reference counterpart: none in Ipopt itself -- MUMPS' own MPI layer behind
IpMumpsSolverInterface.cpp:191-245, SPRAL's multi-GPU knobs IpSpralSolverInterface.cpp:55-67.)"""
import numpy as np
import pytest

import ipopt_amd
from tests.support import kktgen, mirror

SPLIT, ARENA, STATS, TOPRHS, SOL = 0, 1, 2, 3, 4
WHOLE, NOCOLOR = -2, -1


def plans(nranks, subcube, range_local):
    n, r, c, v, _ = kktgen.grid_kkt(40, 36, dof=2, ncon=1, seed=7)
    s = ipopt_amd.KKTSolver(nranks=nranks, subcube=subcube)
    s.initialize_structure(n, r, c, vals=v)
    sym = mirror.fetch(s)
    return n, sym, [s.comm_plan(rk, range_local) for rk in range(nranks)]


def simulate(nranks, plan):
    """(ipopt_amd.multigpu.play_comm_plans: the same check bench.py --gpus N makes in its dry run)"""
    from ipopt_amd import multigpu
    assert len(plan) == nranks
    return multigpu.play_comm_plans(plan)


@pytest.mark.parametrize("nranks", [2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("subcube", [0, 1])
def test_every_rank_issues_a_matching_sequence_of_collectives(nranks, subcube):
    n, sym, plan = plans(nranks, subcube, True)
    own, glo, gsz, gd = sym["owner"], sym["glo"], sym["gsz"], sym["gdepth"]
    # --- the ncclCommSplit calls: a collective over the WHOLE communicator, so every rank makes the same calls (same steps, same order); the colours
    #     partition the ranks of a step into exactly the contiguous ranges [glo, glo + gsz) of that step's replicated fronts ---
    splits = [[(int(d), int(col)) for what, d, col, g, cnt, dt in plan[rk] if what == SPLIT] for rk in range(nranks)]
    steps = [[d for d, col in sp] for sp in splits]
    assert all(st == steps[0] for st in steps), steps
    ranges_of = {}
    for sn in np.nonzero(own < 0)[0]:
        ranges_of.setdefault(int(gd[sn]), set()).add((int(glo[sn]), int(gsz[sn])))
    for j, d in enumerate(steps[0]):
        assert any(g < nranks for _, g in ranges_of[d])           # a split is only made for a step with a range smaller than the machine
        groups = {}
        for rk in range(nranks):
            col = splits[rk][j][1]
            if col != NOCOLOR:
                groups.setdefault(col, []).append(rk)
        # (a whole-machine range of the same step -- the classic top -- has colour = its first rank too; it is never USED for a collective: colour -2)
        assert {(col, len(m)) for col, m in groups.items()} == ranges_of[d], (d, groups, ranges_of[d])
        for col, m in groups.items():
            assert m == list(range(col, col + len(m)))              # contiguous, first rank = colour; key = rank keeps the order
    for d, rs in ranges_of.items():
        if d not in steps[0]:
            assert all(g >= nranks for _, g in rs)                  # no split <=> only the whole machine at that step
    # --- the collectives themselves: playable to the end, every communicator's members agree on (what, step, count, dtype) ---
    done, subcomm = simulate(nranks, plan)
    whole = [rec for rec in plan[0] if rec[0] != SPLIT and rec[2] == WHOLE]
    assert [int(r[0]) for r in whole if r[0] in (STATS, SOL)] == [STATS, SOL]
    assert int(plan[0][-1][4]) == n and all(int(p[-1][0]) == SOL for p in plan)
    # what travels: the arena bytes of the plan equal the arena bytes of the structure (lower triangles of the join fronts, once per range)
    m = np.diff(sym["rowptr"]).astype(np.int64)
    par = sym["parent"]
    join = np.zeros(len(own), dtype=bool)
    for ch in range(len(own)):
        p = par[ch]
        if p >= 0 and own[p] < 0 and not (own[ch] < 0 and (glo[ch], gsz[ch]) == (glo[p], gsz[p])):
            join[p] = True
    seen = {}
    for rk in range(nranks):
        for what, d, col, g, cnt, dt in plan[rk]:
            if what == ARENA:
                seen[(int(d), int(col) if col != WHOLE else 0, int(g))] = int(cnt)
    assert sum(seen.values()) == int((m[join] * (m[join] + 1) // 2).sum())
    if subcube and nranks >= 4:
        assert len(steps[0]) >= 1 and len(subcomm) >= 1             # the mapping really produced a range smaller than the machine


@pytest.mark.parametrize("nranks", [2, 4, 8])
def test_whole_communicator_fall_back_is_one_sum_per_step_on_every_rank(nranks):
    """MI355X_KKT_DISABLE=subcomm / a failed ncclCommSplit: no splits, every rank issues the identical list (zeros for the squares of ranges it is not in)."""
    n, sym, plan = plans(nranks, 1, False)
    ref = [tuple(int(x) for x in rec) for rec in plan[0]]
    assert all(rec[0] != SPLIT and rec[2] == WHOLE for rec in ref)
    for rk in range(1, nranks):
        assert [tuple(int(x) for x in rec) for rec in plan[rk]] == ref
    simulate(nranks, plan)
    # same bytes as the range-local schedule moves in total, but every rank carries all of them
    _, _, plan_local = plans(nranks, 1, True)
    tot_local = {}
    for rk in range(nranks):
        for what, d, col, g, cnt, dt in plan_local[rk]:
            if what == ARENA:
                tot_local[(int(d), int(col), int(g))] = int(cnt)
    assert sum(c for w, d, col, g, c, dt in ref if w == ARENA) == sum(tot_local.values())

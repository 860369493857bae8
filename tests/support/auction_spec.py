"""numpy specification of the DEVICE maximum-product matching scaling (scaling modes 5 / 6; kernels_match.hip.inc follows it step by step).

The job of MC64 (reference knobs `ma97_scaling mc64`, `spral_scaling matching`: IpMa97SolverInterface.cpp:107-190,725-771, IpSpralSolverInterface.cpp:199-204)
by Bertsekas' auction algorithm in its Jacobi form (all free columns bid at once) -- the form that maps onto a GPU -- instead of the successive shortest
paths of matching_scaling.cpp (Duff & Koster), which are serial.

  benefit  b_ij = log|a_ij| - log max_k|a_kj|  <= 0          (column j "person", row i "object", full symmetric pattern)
  a free column j bids for its best row i1 = argmax_i (b_ij - p_i):  bid = p_i1 + min(w1 - w2, LONE) + eps   (w2 = second best value, -inf for a single entry);
  a column whose best value is below -GIVEUP does not bid any more in this phase (deficient pattern: it stays unmatched, the prices stay representable)
  a row takes the highest bid (ties: the smallest column index), its previous owner becomes free, its price becomes the bid
  eps-scaling: phases eps = eps0, eps0 / 4, ... >= eps_final; prices carry over, a phase starts from the assignments that still satisfy eps-complementary slackness

At the end  pi_j = max_i (b_ij - p_i)  (every column, matched or not)  =>  u_i = -p_i, v_j = -pi_j are FEASIBLE duals of the assignment problem with costs
c_ij = -b_ij (u_i + v_j <= c_ij for every entry, by construction), so with r_i = exp(u_i), q_j = exp(v_j) / max_k|a_kj|, s_i = sqrt(r_i q_i):
  |s_i a_ij s_j| <= 1 for EVERY entry (exactly, whatever eps), and r_i |a_ij| q_j >= exp(-eps_final) on the matching.
A column whose rows all went to higher bidders for more than `max_rounds` rounds in a phase stays unmatched (structurally deficient pattern); reported."""
import numpy as np

LONE = 64.0          # largest price increment of one bid (besides eps); the margin of a column with a single entry
GIVEUP = 300.0       # a column whose best value has fallen below -GIVEUP stops bidding: structurally deficient pattern (prices stay representable)


def auction_scaling(n, ptr, idx, absval, eps_final=1.0 / 64, eps0=0.25, max_rounds=8192, phase_rounds=256, stats=None):
    ptr = np.asarray(ptr, dtype=np.int64); idx = np.asarray(idx, dtype=np.int64); a = np.asarray(absval, dtype=np.float64)
    cnt = np.diff(ptr)
    colof = np.repeat(np.arange(n), cnt)
    cmax = np.zeros(n); np.maximum.at(cmax, colof, a)
    empty = ~(cmax > 0)
    with np.errstate(divide="ignore"):
        b = np.where(a > 0, np.log(a) - np.log(np.where(cmax > 0, cmax, 1.0))[colof], -np.inf)
    price = np.zeros(n)
    owner = np.full(n, -1, dtype=np.int64)        # owner[i] = column that holds row i
    mrow = np.full(n, -1, dtype=np.int64)         # mrow[j] = row held by column j
    rounds_total = 0
    eps = eps0
    while True:
        # phase start: keep the pairs that satisfy eps-CS at the current prices, free the rest
        val = b - price[idx]
        w1 = np.full(n, -np.inf); np.maximum.at(w1, colof, val)
        if rounds_total > 0:
            held = mrow >= 0
            hv = np.full(n, -np.inf)
            # value of the held entry of column j
            sel = held[colof] & (idx == mrow[colof])
            hv[colof[sel]] = val[sel]
            drop = held & ~(hv >= w1 - eps)
            owner[mrow[drop]] = -1; mrow[drop] = -1
        gave_up = np.zeros(n, dtype=bool)
        free = np.nonzero((mrow < 0) & ~empty)[0]
        rounds = 0
        cap = max_rounds if eps <= eps_final else phase_rounds
        while free.size and rounds < cap:
            # entries of the free columns
            lens = cnt[free]
            start = ptr[free]
            tot = int(lens.sum())
            seg = np.repeat(np.arange(free.size), lens)
            off = np.arange(tot) - np.repeat(np.cumsum(lens) - lens, lens)
            e = np.repeat(start, lens) + off
            v = b[e] - price[idx[e]]
            # best and second best per free column (ties on the value: the entry that comes first in the column)
            best = np.full(free.size, -np.inf); np.maximum.at(best, seg, v)
            isb = v == best[seg]
            first = np.full(free.size, np.iinfo(np.int64).max); np.minimum.at(first, seg[isb], e[isb])
            v2 = np.where(e == first[seg], -np.inf, v)
            second = np.full(free.size, -np.inf); np.maximum.at(second, seg, v2)
            ok = best >= -GIVEUP
            gave_up[free[~ok & np.isfinite(best)]] = True
            with np.errstate(invalid="ignore"):
                inc = np.minimum(best - second, LONE)
            fj = free[ok]; i1 = idx[first[ok]]; bid = price[i1] + inc[ok] + eps
            # every row takes its highest bid, ties -> smallest column
            top = np.full(n, -np.inf); np.maximum.at(top, i1, bid)
            win = bid == top[i1]
            wcol = np.full(n, np.iinfo(np.int64).max); np.minimum.at(wcol, i1[win], fj[win])
            rows = np.nonzero(np.isfinite(top))[0]
            old = owner[rows]
            mrow[old[old >= 0]] = -1
            owner[rows] = wcol[rows]; mrow[wcol[rows]] = rows; price[rows] = top[rows]
            free = np.nonzero((mrow < 0) & ~empty & ~gave_up)[0]
            rounds += 1
        rounds_total += rounds
        if stats is not None: stats.setdefault("phases", []).append((eps, rounds, int(free.size)))
        if eps <= eps_final: break
        eps = max(eps / 4.0, eps_final)
    val = b - price[idx]
    pi = np.full(n, -np.inf); np.maximum.at(pi, colof, val)
    pi = np.where(np.isfinite(pi), pi, 0.0)
    u = -price; v = -pi
    logq = np.where(empty, 0.0, v - np.log(np.where(cmax > 0, cmax, 1.0)))
    s = np.exp(0.5 * (u + logq))
    s = np.where(np.isfinite(s) & (s > 0), s, 1.0)
    unmatched = int(((mrow < 0)).sum())
    if stats is not None: stats["rounds"] = rounds_total; stats["mrow"] = mrow
    return s, unmatched

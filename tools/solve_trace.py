"""Reads the file of MI355X_KKT_TRACE=solve=<file> (4 wall-clock stamps per workgroup of the data-flow solve sweeps, 10 ns units) and prints, per chain:
when its first link started waiting, when its rows had their start value, when the first / last link published, when the rows beyond were done."""
import collections, sys
rows = [l.split() for l in open(sys.argv[1])]
for d in "FB":
    R = [(int(c), int(w), int(nl), int(t), [int(x) for x in ts]) for dd, c, w, nl, t, *ts in rows if dd == d]
    if not R: continue
    t0 = min(r[4][0] for r in R); t1 = max(r[4][3] for r in R)
    print(("forward" if d == "F" else "backward") + f" sweep: {len(R)} workgroups, {(t1 - t0) / 100:.1f} us")
    ch = collections.defaultdict(list)
    for r in R: ch[r[0]].append(r)
    f = lambda x: (x - t0) / 100.0
    print("  chain links tail | first wg in   start value   first link out   last link out   tails out | us per hop")
    order = sorted(ch, reverse=True)
    for c in order[: int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
        L = sorted([r for r in ch[c] if r[1] >= 0], key=lambda r: r[1]); nl = L[0][2]
        links, tails = L[:nl], L[nl:]
        if d == "B": tails = [r for r in ch[c] if r[1] < 0]          # (the dot workgroups: partial sums over the rows beyond the chain)
        first, last = (links[0], links[-1]) if d == "F" else (links[0], links[-1])
        print(f"  {c:5d} {nl:5d} {L[0][3]:5d} | {f(min(x[4][0] for x in L)):8.1f} {f(first[4][1]):10.1f} {f(first[4][3]):12.1f} {f(last[4][3]):14.1f} "
              f"{(max(f(t[4][3]) for t in tails) if tails else float('nan')):12.1f} | {(f(last[4][3]) - f(first[4][3])) / max(nl - 1, 1):6.2f}")

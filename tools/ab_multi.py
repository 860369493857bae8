"""Development aid: A/B/C... of environment switches on one bench workload in ONE process and ONE gpu call (boxes differ by several per cent;
the workload is generated and torch imported once).  Every variant gets a fresh handle (the switches are read when a handle is made / set up),
`reps` factorisations + solves (best device time of each), the per-kernel-kind times of profile(), and its solution compared bitwise with variant 0's.
usage: tools/ab_multi.py WORKLOAD REPS "" "ENV=1" "ENV=1 OTHER=2" ...        (an empty string = no switch)"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ipopt_amd, bench

wl, reps, variants = sys.argv[1], int(sys.argv[2]), sys.argv[3:] or [""]
n, r, c, v, neg = bench.make_workload(wl)
dv = torch.tensor(v, dtype=torch.float64, device="cuda"); db = torch.ones(n, dtype=torch.float64, device="cuda")
ref = None
rounds = int(os.environ.get("AB_ROUNDS", "2"))        # the variants are gone through `rounds` times (drift of the box's clocks shows as a difference between the rounds)
best = {}
for rd in range(rounds):
    for vi, var in enumerate(variants):
        keys = []
        for kv in var.split():
            k, _, val = kv.partition("="); os.environ[k] = val or "1"; keys.append(k)
        s = ipopt_amd.KKTSolver(device=0)
        s.initialize_structure(n, r, c, vals=v)
        dx = torch.empty_like(db)
        tf, ts = [], []
        for i in range(reps):
            st = s.factor_device(dv.data_ptr()); tf.append(s.info().time_factor_ms)
            s.solve_device2(db.data_ptr(), dx.data_ptr()); ts.append(s.info().time_solve_ms)
        x = dx.cpu().numpy()
        I = s.info()
        if ref is None: ref = x.copy()
        rec = dict(factor_ms=round(min(tf), 3), solve_ms=round(min(ts), 3), status=int(st[0]), neg_ok=bool(I.num_neg == neg), bitwise=bool(np.array_equal(x, ref)),
                   maxdiff=float(np.abs(x - ref).max()))
        if rd == rounds - 1:
            prof = s.profile(3)
            rec["by_kind"] = {k: round(ms / 3, 3) for k, (ms, ln) in prof.items() if ln}
        print(f"round {rd} [{var or 'baseline'}] " + json.dumps(rec), flush=True)
        b = best.setdefault(var, rec)
        if rec["factor_ms"] < b["factor_ms"]: best[var] = rec
        del s
        for k in keys: os.environ.pop(k, None)
print("best factor_ms: " + ", ".join(f"[{k or 'baseline'}] {b['factor_ms']}" for k, b in best.items()))

"""What one delayed-pivot edit costs: wall time of  mi355x_kkt_delay_columns  (host: restructure_delays + the re-analysis of everything
that follows from the permutation; device: re-setup of what depends on the structure) + the refactorisation behind it, against one factorisation.
usage: python tools/delay_cost.py [workload] [columns to delay] -- with a GPU: the whole cycle; without: the host part only (numeric set-up skipped)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ipopt_amd, bench
wl = sys.argv[1] if len(sys.argv) > 1 else "synth_1e6"
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 100
verbose = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n, r, c, v, neg = bench.make_workload(wl)
try:
    import torch
    gpu = torch.cuda.is_available()
except Exception:
    gpu = False
s = ipopt_amd.KKTSolver(device=0 if gpu else -1, delay_rounds=0, verbose=verbose)
t0 = time.perf_counter(); s.initialize_structure(n, r, c, vals=v); t_an = time.perf_counter() - t0
I0 = s.info()
out = {"workload": wl, "kkt_dim": n, "analyse_s": I0.time_analyse, "initialize_structure_wall_s": t_an, "nnz_L": I0.nnz_l, "supernodes": I0.num_sn, "gpu": gpu}
tf = None
if gpu:
    dv = torch.tensor(v, dtype=torch.float64, device="cuda")
    for _ in range(3):
        s.factor_device(dv.data_ptr())
    tf = min(s.info().time_factor_ms for _ in range(1))
    ts = []
    for _ in range(5):
        s.factor_device(dv.data_ptr()); ts.append(s.info().time_factor_ms)
    tf = min(ts); out["factor_ms"] = tf
rng = np.random.default_rng(7)
rounds = []
for rd in range(3):
    cols = rng.choice(n, size=ncol, replace=False) + 1            # (caller's numbering, 1-based: random columns all over the tree)
    t0 = time.perf_counter(); moved = s.delay_columns(cols); t_edit = time.perf_counter() - t0
    row = {"columns": ncol, "moved": moved, "edit_wall_ms": 1e3 * t_edit}
    if gpu:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st = s.refactor() if hasattr(s, "refactor") else s.factor_device(dv.data_ptr())
        torch.cuda.synchronize(); row["first_refactor_wall_ms"] = 1e3 * (time.perf_counter() - t0)      # (graphs are captured again)
        s.factor_device(dv.data_ptr()); row["factor_ms_after"] = s.info().time_factor_ms
        row["edit_over_factor"] = row["edit_wall_ms"] / tf
        row["whole_cycle_over_factor"] = (row["edit_wall_ms"] + row["first_refactor_wall_ms"]) / tf
    I = s.info(); row["nnz_L"] = I.nnz_l
    rounds.append(row)
out["rounds"] = rounds
print(json.dumps(out))

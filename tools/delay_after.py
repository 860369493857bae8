"""factorisation time before / after delayed-pivot edits of 100 random columns each (what an edit does to the SCHEDULE): python tools/delay_after.py [workload | g:NX:NY]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ipopt_amd, bench
from tests.support import kktgen
for wl in (sys.argv[1:] or ["g:250:160", "synth_1e6"]):
    n, r, c, v, neg = bench.make_workload(wl) if not wl.startswith("g:") else kktgen.grid_kkt(int(wl.split(":")[1]), int(wl.split(":")[2]), dof=3, ncon=2, seed=77, sigma_exp=6.0)
    s = ipopt_amd.KKTSolver(delay_rounds=0); s.initialize_structure(n, r, c, vals=v)
    dv = torch.tensor(v, dtype=torch.float64, device="cuda")
    def tf():
        for _ in range(2): s.factor_device(dv.data_ptr())
        return min((s.factor_device(dv.data_ptr()), s.info().time_factor_ms)[1] for _ in range(5))
    I = s.info(); out = {"workload": wl, "kkt_dim": n, "factor_ms": [tf()], "supernodes": [I.num_sn], "levels": [I.num_levels]}
    rng = np.random.default_rng(7)
    for rd in range(3):
        s.delay_columns(rng.choice(n, size=100, replace=False) + 1)
        st = s.factor_device(dv.data_ptr()); assert st[0] == 0 and st[1] == neg
        I = s.info(); out["factor_ms"].append(tf()); out["supernodes"].append(I.num_sn); out["levels"].append(I.num_levels)
    print(json.dumps(out), flush=True)

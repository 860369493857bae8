"""factorisation device time of a workload under the current environment (A/B of MI355X_KKT_DISABLE / _TUNE settings from the shell): python tools/factor_time.py <workload | npz:path> [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ipopt_amd, bench
wl = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n, r, c, v, neg = bench.make_workload(wl)
s = ipopt_amd.KKTSolver(device=0); s.initialize_structure(n, r, c, vals=v)
dv = torch.tensor(v, dtype=torch.float64, device="cuda")
ts = []
for i in range(reps + 2):
    st = s.factor_device(dv.data_ptr()); ts.append(s.info().time_factor_ms)
assert st[0] == 0 and st[1] == neg
I = s.info()
print(f"{os.path.basename(wl)} DISABLE={os.environ.get('MI355X_KKT_DISABLE','')} TUNE={os.environ.get('MI355X_KKT_TUNE','')}: factor {min(ts[2:]):.2f} ms = {I.flops_factor / min(ts[2:]) / 1e9:.1f} TFLOP/s")

"""Development aid: analyse once, then a few factorisations + solves of one bench workload (for rocprofv3 timelines)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ipopt_amd, bench
wl = sys.argv[1] if len(sys.argv) > 1 else "synth_1e6"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n, r, c, v, neg = bench.make_workload(wl)
s = ipopt_amd.KKTSolver(device=0, verbose=1)
s.initialize_structure(n, r, c, vals=v)
dv = torch.tensor(v, dtype=torch.float64, device="cuda"); db = torch.ones(n, dtype=torch.float64, device="cuda"); dx = torch.empty_like(db)
torch.cuda.synchronize()
for i in range(reps):
    t0 = time.perf_counter()
    st = s.factor_device(dv.data_ptr())
    t1 = time.perf_counter()
    s.solve_device2(db.data_ptr(), dx.data_ptr())
    print(f"rep {i}: status {st} factor wall {1e3 * (t1 - t0):.2f} ms (device {s.info().time_factor_ms:.2f}) solve device {s.info().time_solve_ms:.2f} ms", flush=True)
I = s.info()
x = dx.cpu().numpy()
from tests.support import kktgen
K = kktgen.to_scipy(n, r, c, v)
res = np.abs(K @ x - 1.0).max() / (abs(K).sum(axis=1).max() * np.abs(x).max() + 1.0)
print(f"num_neg {I.num_neg} (expected {neg}) num_two {I.num_two} num_delay {I.num_small} u_sensitive {I.u_sensitive} big fronts {I.num_big_fronts} fast pivot blocks {I.num_fast_blocks} scaled residual {res:.2e}", flush=True)

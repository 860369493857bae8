"""Development aid: one bench workload under several option sets, e.g.  python tools/opt_sweep.py lukvle1_1e6 leaf_cols=32 nd_leaf=64,nemin=16
(every argument after the workload is one comma-separated option set; '-' = defaults).  Prints factor / solve device ms, nnz(L), residual."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ipopt_amd, bench
from tests.support import kktgen
wl = sys.argv[1]
n, r, c, v, neg = bench.make_workload(wl)
K = kktgen.to_scipy(n, r, c, v)
dv = torch.tensor(v, dtype=torch.float64, device="cuda"); db = torch.ones(n, dtype=torch.float64, device="cuda"); dx = torch.empty_like(db)
for spec in sys.argv[2:]:
    opts = {} if spec == "-" else {k: (float(x) if "." in x or "e" in x else int(x)) for k, x in (kv.split("=") for kv in spec.split(","))}
    s = ipopt_amd.KKTSolver(device=0, **opts)
    t0 = time.perf_counter(); s.initialize_structure(n, r, c, vals=v); ta = time.perf_counter() - t0
    tf, ts = [], []
    for i in range(6):
        st = s.factor_device(dv.data_ptr()); s.solve_device2(db.data_ptr(), dx.data_ptr())
        tf.append(s.info().time_factor_ms); ts.append(s.info().time_solve_ms)
    torch.cuda.synchronize()
    I = s.info(); x = dx.cpu().numpy()
    res = np.abs(K @ x - 1.0).max() / (abs(K).sum(axis=1).max() * np.abs(x).max() + 1.0)
    print(f"[{spec}] factor {min(tf):.3f} ms solve {min(ts):.3f} ms step {min(tf) + 2 * min(ts):.3f} | analyse {ta:.2f} s nsn {I.num_sn} levels {I.num_levels} nnzL {I.nnz_l} maxfront {I.maxfront} neg {I.num_neg}/{neg} res {res:.1e}", flush=True)
    del s

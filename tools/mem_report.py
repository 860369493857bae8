"""Pool sizes and peak device memory of a workload: cb_doubles, L + contribution-block pool, device memory in use before / after
set-up and after a factor + solve, factor / solve device ms, residual.  usage: python tools/mem_report.py <workload | npz:path> [...]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ipopt_amd, bench
from tests.support import kktgen
for wl in sys.argv[1:]:
    n, r, c, v, neg = bench.make_workload(wl)
    K = kktgen.to_scipy(n, r, c, v)
    free0, total = torch.cuda.mem_get_info()
    s = ipopt_amd.KKTSolver(device=0)
    t0 = time.perf_counter(); s.initialize_structure(n, r, c, vals=v); t_init = time.perf_counter() - t0
    I = s.info()
    colptr = s.symbolic(1, I.num_sn + 1).astype(np.int64); rowptr = s.symbolic(2, I.num_sn + 1).astype(np.int64)
    k = np.diff(colptr); m = np.diff(rowptr)
    free1, _ = torch.cuda.mem_get_info()
    dv = torch.tensor(v, dtype=torch.float64, device="cuda"); b = K @ np.ones(n); db = torch.tensor(b, dtype=torch.float64, device="cuda"); dx = torch.empty_like(db)
    out = {"workload": wl, "kkt_dim": n, "nnz": int(len(v)), "supernodes": I.num_sn, "levels": I.num_levels, "maxfront": I.maxfront, "nnz_L": I.nnz_l, "flops_per_factor": I.flops_factor,
           "analyse_s": I.time_analyse, "initialize_structure_wall_s": t_init, "cb_doubles": I.cb_doubles, "cb_GiB": I.cb_doubles * 8 / 2**30,
           "panels_GiB": float((m * k).sum() * 8 / 2**30), "cb_if_every_block_were_separate_GiB": float(((m - k) ** 2).sum() * 8 / 2**30),
           "device_total_GiB": total / 2**30, "device_used_by_setup_GiB": (free0 - free1) / 2**30}
    pl = s.symbolic(27, 5).astype(np.int64)      # the storage plan of the contribution blocks (symbolic.cpp step 12a)
    out["cb_plan"] = {"window_levels": int(pl[0]), "every_block_resident_GiB": float(((pl[1] & 0xffffffff) | (pl[2] << 32)) * 8 / 2**30),
                      "never_reused_GiB": float(((pl[3] & 0xffffffff) | (pl[4] << 32)) * 8 / 2**30)}
    try:
        tf, ts = [], []
        for _ in range(3):
            st = s.factor_device(dv.data_ptr()); tf.append(s.info().time_factor_ms)
            s.solve_device2(db.data_ptr(), dx.data_ptr()); ts.append(s.info().time_solve_ms)
        free2, _ = torch.cuda.mem_get_info()
        x = dx.cpu().numpy()
        res = float(np.abs(K @ x - b).max() / (abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max()))
        J = s.info()
        out.update(status=st[0], num_neg=st[1], num_neg_expected=neg, factor_ms=min(tf), solve_ms=min(ts), factor_TFLOPs=I.flops_factor / min(tf) / 1e9, scaled_residual=res,
                   device_used_peak_GiB=(free0 - free2) / 2**30, num_small=J.num_small, num_delayed=J.num_delayed)
    except Exception as e:
        out["error"] = str(e)[:300]
    print(json.dumps(out), flush=True)
    del s, dv, db, dx
    torch.cuda.empty_cache()

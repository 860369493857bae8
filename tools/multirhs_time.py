"""nrhs right-hand sides in ONE call against nrhs single solves (the solves of several columns in flight at once, one context each --
numeric.hip, solve contexts): device time of mi355x_kkt_solve_device with nrhs = 1, 2, 4, 8 on a bench workload, bitwise comparison with single solves.
usage: python tools/multirhs_time.py [workload ...]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ipopt_amd, bench
for wl in (sys.argv[1:] or ["synth_1e6"]):
    n, r, c, v, neg = bench.make_workload(wl)
    s = ipopt_amd.KKTSolver(device=0)
    s.initialize_structure(n, r, c, vals=v)
    dv = torch.tensor(v, dtype=torch.float64, device="cuda")
    st = s.factor_device(dv.data_ptr())
    assert st[0] == 0 and st[1] == neg
    B = torch.tensor(np.random.default_rng(3).standard_normal((8, n)), dtype=torch.float64, device="cuda")
    out = {"workload": wl, "kkt_dim": n}
    singles = torch.empty_like(B)
    for q in range(8):
        s.solve_device2(B[q].data_ptr(), singles[q].data_ptr())
    torch.cuda.synchronize()
    for mode in ("default_measured_choice", "contexts_forced", "one_after_the_other"):
        os.environ.pop("MI355X_KKT_TUNE", None)
        if mode == "contexts_forced":
            os.environ["MI355X_KKT_TUNE"] = "solve_ctx_force=1"
        if mode == "one_after_the_other":
            os.environ["MI355X_KKT_DISABLE"] = "solve_ctx"
        res = {}
        for nrhs in (1, 2, 4, 8):
            X = torch.empty((nrhs, n), dtype=torch.float64, device="cuda")
            for _ in range(2):
                s.solve_device2(B.data_ptr(), X.data_ptr(), nrhs=nrhs)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                s.solve_device2(B.data_ptr(), X.data_ptr(), nrhs=nrhs)
            torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / reps
            res[str(nrhs)] = {"ms": ms, "bitwise_equal_to_single_solves": bool(torch.equal(X, singles[:nrhs]))}
        res["nrhs8_over_nrhs1"] = res["8"]["ms"] / res["1"]["ms"]
        out[mode] = res
    os.environ.pop("MI355X_KKT_DISABLE", None); os.environ.pop("MI355X_KKT_TUNE", None)
    print(json.dumps(out), flush=True)

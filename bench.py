#!/usr/bin/env python
"""bench.py -- KKT factor+solve throughput of the MI355X backend (BASELINE.json metric).

A "step" = the linear-algebra work of one Ipopt iteration on one KKT system: ONE numeric
factorisation (with inertia) + TWO triangular solves (the solve and the forced refinement step,
reference IpPDFullSpaceSolver.cpp:40-47,256-346), values and right-hand side already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

Workloads (synthetic data of the named shape, inertia known by construction -- tests/support/kktgen.py):
    lukvle1_1e4   BASELINE.json configs[1]: KKT of ScalableProblems LukVlE1 n=10 000 (dim 19 998, triplet nnz 69 991)
    lukvle1_1e6   the 10^6-variable target instance (dim 1 999 998)
    grid_1e5      PDE-constrained-like KKT, 160x125 grid, 3 dof + 2 constraints per node (CI-sized sibling of configs[3])
    synth_1e6     BASELINE.json configs[3]: n = 10^6, nnz ~ 2e7 (500x400 grid, 3 dof + 2 constraints per node)
Default: synth_1e6 at every N -- BASELINE.json quotes the GFLOP/s metric on the synthetic n = 10^6 system "at 1, 2, 4 and 8
GPUs"; it fits one GPU, and the driver's scaling efficiency needs the same workload at every N.  The N=1 line also carries
the LukVlE1 configurations (configs[1] and the 10^6-variable target) under "also".

Prints ONE JSON line (rank 0).  value = algorithmic GFLOP/s of factor + 2 solves, flop counts as defined
in SURVEY 8(d): F_fact = sum_j (c_j-1)(c_j+2), F_solve = 4 nnz(L) - 3 n per rhs, for the ordering actually used.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F64_PEAK_TFLOPS = 78.6    # AMD public MI355X spec, fp64 matrix (v_mfma_f64_16x16x4_f64); see DESIGN.md
MFMA_F64_SUSTAINED_TFLOPS = 47.6   # measured on the box: tools/micro/mfma_f64_peak.hip (back-to-back independent MFMAs, >= 2 waves/SIMD)


def make_workload(name):
    from tests.support import kktgen
    if name == "lukvle1_1e4":
        return kktgen.lukvl_like(10_000, seed=20260923)
    if name == "lukvle1_1e6":
        return kktgen.lukvl_like(1_000_000, seed=20260923)
    if name == "grid_1e5":
        return kktgen.grid_kkt(160, 125, dof=3, ncon=2, seed=20260923, sigma_exp=8.0)
    if name == "synth_1e6":
        return kktgen.grid_kkt(500, 400, dof=3, ncon=2, seed=20260923, sigma_exp=8.0)
    raise SystemExit(f"unknown workload {name}")


def per_kind_work(solver):
    """algorithmic bytes / flops per kernel kind for ONE factorisation + ONE solve (SURVEY 8(d) per-unit figures
    summed over the fronts each kernel kind processes)."""
    I = solver.info()
    g = solver.symbolic
    colptr, rowptr, acolptr = g(1, I.num_sn + 1).astype(np.int64), g(2, I.num_sn + 1).astype(np.int64), g(7, I.n + 1).astype(np.int64)
    k = np.diff(colptr); m = np.diff(rowptr)
    nnza = acolptr[colptr[1:]] - acolptr[colptr[:-1]]
    nnzl = k * m - k * (k - 1) // 2
    cls = np.where(m <= 32, 0, np.where(m <= 64, 1, np.where(m <= 128, 2, 3)))
    bytes_f = 12 * nnza + 8 * nnzl + 4 * m                       # B_fact per front
    bytes_s = 8 * nnzl + 4 * m + 12 * k                          # one sweep (half of B_solve) per front
    mu = m - k
    out = {}
    names_f = ["front_wave", "front_lds64", "front_lds128"]
    for c in range(3):
        out[names_f[c]] = dict(bytes=int(bytes_f[cls == c].sum()), flops=0)
    big = cls == 3
    out["big_schur"] = dict(bytes=int((16 * (mu[big] * (mu[big] + 1) // 2) + 16 * mu[big] * k[big]).sum()), flops=int((mu[big] * (mu[big] + 1) * k[big]).sum()))
    out["big_assemble"] = dict(bytes=int((12 * nnza[big] + 8 * (m[big] * k[big] + mu[big] * (mu[big] + 1) // 2)).sum()), flops=0)
    out["big_trsm"] = dict(bytes=int((24 * mu[big] * k[big]).sum()), flops=int((mu[big] * k[big] * k[big]).sum()))
    out["big_diag"] = dict(bytes=int((16 * k[big] * k[big]).sum()), flops=int((k[big] ** 3 // 3).sum()))
    out["fwd_wave"] = out["bwd_wave"] = dict(bytes=int(bytes_s[cls == 0].sum()), flops=0)
    out["fwd_lds"] = out["bwd_lds"] = dict(bytes=int(bytes_s[(cls == 1) | (cls == 2)].sum()), flops=0)
    out["fwd_big"] = out["bwd_big"] = dict(bytes=int((8 * k[big] * k[big] + 12 * k[big] + 4 * m[big]).sum()), flops=0)          # pivot blocks (stored inverse)
    out["fwd_big_upd"] = out["bwd_big_dot"] = dict(bytes=int((8 * mu[big] * k[big] + 12 * mu[big]).sum()), flops=0)            # rows below them
    out["gather_scale"] = dict(bytes=int(8 * I.nnz_in + 4 * I.nnz_in + 8 * I.nnz_a * 9), flops=0)
    out["solve_perm"] = dict(bytes=int(2 * 28 * I.n), flops=0)
    out["stats"] = dict(bytes=16 * I.num_sn, flops=0)
    return out


def cpu_baseline(n, r, c, v, b, x_gpu, nsolve):
    """the reference's own CPU path (TripletToCSRConverter + PardisoMKLSolverInterface, oneMKL PARDISO), prebuilt in
    oracle/_ref by oracle/ref_build.mk, timed on this box's host cores; falls back to the C oracle port."""
    tool = os.path.join(ROOT, "oracle", "_ref", "ref_kkt_solve")
    if os.path.exists(tool):
        with tempfile.NamedTemporaryFile(suffix=".kkt", delete=False) as f:
            f.write(np.array([n, len(v)], dtype=np.int32).tobytes()); f.write(r.astype(np.int32).tobytes())
            f.write(c.astype(np.int32).tobytes()); f.write(v.astype(np.float64).tobytes()); f.write(b.astype(np.float64).tobytes())
            path = f.name
        best = None
        ncores = os.cpu_count() or 1
        for threads in sorted({min(16, ncores)} if len(v) > 10_000_000 else {1, min(16, ncores)}):   # (1 thread on the 2e7-entry system alone would take minutes)
            env = dict(os.environ, MKL_NUM_THREADS=str(threads), OMP_NUM_THREADS=str(threads), MKL_DYNAMIC="FALSE")
            nfac = 4 if n <= 300_000 else (3 if len(v) <= 10_000_000 else 2)
            try:
                out = subprocess.run([tool, path, str(nfac), str(nsolve)], capture_output=True, text=True, env=env, timeout=900).stdout
                j = json.loads(out.strip().splitlines()[-1])
            except Exception:
                continue
            t = j["factor_plus_first_solve_s"] + j["extra_solves_s"]
            if best is None or t < best[0]:
                best = (t, threads, j, nfac)
        os.unlink(path)
        if best is not None:
            t, threads, j, nfac = best
            return dict(seconds_per_step=t, cores=threads, kind="reference", num_neg=j["num_neg"],
                        sample=f"same KKT system, {nfac - 1} timed factor+{nsolve}-solve steps after one warm-up (symbolic excluded), "
                               f"reference PardisoMKLSolverInterface on oneMKL PARDISO, best of MKL_NUM_THREADS in {{1,{min(16, ncores)}}} (16 only for the 2e7-entry system)")
    # port: the C oracle (scalar, 1 core) on a bounded sample (leading principal sub-band of the workload if it is large)
    from oracle import kkt_oracle as ko
    t0 = time.perf_counter(); ko.factor_solve(n, r, c, v, np.stack([b] * nsolve), u=1e-8); t = time.perf_counter() - t0
    return dict(seconds_per_step=t, cores=1, kind="port", num_neg=None, sample="same KKT system, one factor+solves with oracle/ldlt_oracle.c")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1 or os.environ.get("MI355X_KKT_FORCE_MULTI"):
        from ipopt_amd import multigpu
        return multigpu.bench_main(args, rank, world, local)

    import torch
    import ipopt_amd
    wl = "synth_1e6" if args.workload == "auto" else args.workload
    torch.cuda.set_device(0)
    n, r, c, v, neg = make_workload(wl)
    from tests.support import kktgen
    K = kktgen.to_scipy(n, r, c, v)
    b = K @ np.ones(n)
    s = ipopt_amd.KKTSolver(device=0)
    s.initialize_structure(n, r, c, vals=v)
    I = s.info()
    dv = torch.tensor(v, dtype=torch.float64, device="cuda")
    db = torch.tensor(b, dtype=torch.float64, device="cuda")
    dx = torch.empty_like(db)
    torch.cuda.synchronize()   # inputs complete before the library's own stream touches them
    NSOLVE = 2

    def step():
        st, nneg, nzero = s.factor_device(dv.data_ptr())
        for _ in range(NSOLVE):
            s.solve_device2(db.data_ptr(), dx.data_ptr())
        return st, nneg

    for _ in range(max(args.warmup, 1)):
        st, nneg = step()
    assert st == 0 and nneg == neg, f"inertia {nneg} != {neg} (status {st})"
    x = dx.cpu().numpy()
    res = float(np.abs(K @ x - b).max() / (abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max()))
    assert res <= 1e-12, f"scaled residual {res}"
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    flops_step = I.flops_factor + NSOLVE * I.flops_solve
    bytes_step = I.bytes_factor + NSOLVE * I.bytes_solve
    J = s.info()

    # ---- roofline of the dominant kernel: hip events around every launch (eager), averaged over reps ----
    reps = 5
    prof = s.profile(reps)
    work = per_kind_work(s)
    per_rep = {kname: (ms / reps, ln // reps) for kname, (ms, ln) in prof.items() if ln > 0}
    # the solve kinds run once per profile rep; a step has NSOLVE solves
    weight = {kname: (NSOLVE if (kname.startswith("fwd") or kname.startswith("bwd") or kname == "solve_perm") else 1) for kname in per_rep}
    dom = max(per_rep, key=lambda kname: per_rep[kname][0] * weight[kname])
    dms, dlaunch = per_rep[dom]
    w = work.get(dom, dict(bytes=0, flops=0))
    if dom == "big_schur":
        ach = w["flops"] / (dms * 1e-3) / 1e12
        roof = dict(bound="mfma", kernel="k_big_schur", achieved=ach, peak=MFMA_F64_PEAK_TFLOPS, unit="TFLOP/s", frac=ach / MFMA_F64_PEAK_TFLOPS,
                    peak_sustained_measured=MFMA_F64_SUSTAINED_TFLOPS, frac_of_sustained=ach / MFMA_F64_SUSTAINED_TFLOPS)
    else:
        ach = w["bytes"] / (dms * 1e-3) / 1e9
        kn = {"front_wave": "k_front_reg<64,4>", "front_lds64": "k_front_reg<64,8>", "front_lds128": "k_front_reg<256,6|8>", "big_diag": "k_big_diag_reg<4>",
              "fwd_wave": "k_fwd<64,false>", "bwd_wave": "k_bwd<64,false>", "fwd_lds": "k_fwd<*,false>", "bwd_lds": "k_bwd<*,false>",
              "fwd_big": "k_fwd_grp", "bwd_big": "k_bwd_grp", "fwd_big_upd": "k_fwd_grp_upd", "bwd_big_dot": "k_bwd_grp_dot"}.get(dom, "k_" + dom)
        roof = dict(bound="hbm", kernel=kn, achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS)
    roof.update(launches_per_factor_solve=dlaunch, avg_launch_us=1e3 * dms / max(dlaunch, 1),
                algorithmic_bytes_per_launch=w["bytes"] / max(dlaunch, 1), algorithmic_flops_per_launch=w["flops"] / max(dlaunch, 1), traffic=None)
    tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            if tj.get("workload") == wl and tj.get("kernel") == roof["kernel"]:
                roof["traffic"] = tj["hbm_bytes_per_factorisation"] / max(dlaunch, 1)     # per launch, like `achieved`
        except Exception:
            pass
    kernel_ms = {kname: round(ms * weight[kname], 4) for kname, (ms, _) in per_rep.items()}

    line = {
        "metric": "KKT factor+solve GFLOP/s (1 numeric LDL^T factorisation + 2 solves per Ipopt iteration)",
        "value": flops_step / dt / 1e9, "unit": "GFLOP/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": wl, "kkt_dim": n, "triplet_nnz": int(len(v)), "nnz_L": I.nnz_l, "flops_per_factor": I.flops_factor,
                   "flops_per_solve": I.flops_solve, "solves_per_step": NSOLVE, "ordering": "nested dissection + minimum-degree leaves (own)",
                   "supernodes": I.num_sn, "tree_levels": I.num_levels, "maxfront": I.maxfront, "num_neg": nneg, "scaled_residual": res},
        "algorithmic_GBps": bytes_step / dt / 1e9,
        "device_ms": {"factor": J.time_factor_ms, "solve": J.time_solve_ms, "by_kernel_per_step": kernel_ms},
        "analyse_s": I.time_analyse,
        "roofline": roof,
    }
    if args.workload == "auto" and not args.no_also:
        # the other single-GPU configurations (same step definition, GPU only); the default line stays on the metric's config
        also = {}
        del s, dv, db, dx
        for w2 in ("lukvle1_1e4", "lukvle1_1e6"):
            try:
                n2, r2, c2, v2, neg2 = make_workload(w2)
                s2 = ipopt_amd.KKTSolver(device=0); s2.initialize_structure(n2, r2, c2, vals=v2)
                K2 = kktgen.to_scipy(n2, r2, c2, v2); b2 = K2 @ np.ones(n2)
                dv2 = torch.tensor(v2, dtype=torch.float64, device="cuda"); db2 = torch.tensor(b2, dtype=torch.float64, device="cuda"); dx2 = torch.empty_like(db2)
                torch.cuda.synchronize()
                def step2():
                    st_ = s2.factor_device(dv2.data_ptr())
                    for _ in range(NSOLVE):
                        s2.solve_device2(db2.data_ptr(), dx2.data_ptr())
                    return st_
                for _ in range(3):
                    st2 = step2()
                torch.cuda.synchronize(); t1 = time.perf_counter()
                for _ in range(20):
                    step2()
                torch.cuda.synchronize(); dt2 = (time.perf_counter() - t1) / 20
                x2 = dx2.cpu().numpy(); I2 = s2.info()
                res2 = float(np.abs(K2 @ x2 - b2).max() / (abs(K2).sum(axis=1).max() * np.abs(x2).max() + np.abs(b2).max()))
                also[w2] = {"kkt_dim": n2, "ms_per_step": dt2 * 1e3, "GFLOP/s": (I2.flops_factor + NSOLVE * I2.flops_solve) / dt2 / 1e9,
                            "algorithmic_GBps": (I2.bytes_factor + NSOLVE * I2.bytes_solve) / dt2 / 1e9, "num_neg_ok": bool(st2[0] == 0 and st2[1] == neg2),
                            "scaled_residual": res2, "device_ms": {"factor": I2.time_factor_ms, "solve": I2.time_solve_ms}}
                del s2, dv2, db2, dx2
            except Exception as e:      # never lose the main line over the extras
                also[w2] = {"error": str(e)[:200]}
        line["also"] = also
    if not args.no_cpu_baseline:
        cb = cpu_baseline(n, r, c, v, b, x, NSOLVE)
        line["cpu_baseline"] = {"value": flops_step / cb["seconds_per_step"] / 1e9, "unit": "GFLOP/s", "cores": cb["cores"], "kind": cb["kind"],
                                "sample": cb["sample"], "ms_per_step": cb["seconds_per_step"] * 1e3, "host_cores_available": os.cpu_count()}
    print(json.dumps(line))


if __name__ == "__main__":
    main()

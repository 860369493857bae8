"""-m gpu: end-to-end -- the UNMODIFIED reference (oracle/_ref/libipopt_ref.so) drives the MI355X backend
through its own plug-in point and must reproduce the iteration sequence of its CPU run (golden
*.iters from tests/golden/make_golden.sh): same iteration count, same per-iteration
objective / inf_pr / inf_du / lg(mu) / lg(rg) / #line-search columns (SURVEY 8(c) pin 4)."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ipopt_mi355x_driver")


def run_driver(problem, n, solver="mi355x"):
    env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1")   # Ipopt's own BLAS-1: one host thread
    out = subprocess.run([DRIVER, problem, str(n), "--solver", solver], capture_output=True, text=True, timeout=900, cwd="/tmp", env=env).stdout
    iters = []
    for ln in out.splitlines():
        f = ln.split()
        if len(f) >= 10 and f[0].rstrip("r").isdigit() and ln.startswith(" "):
            iters.append(" ".join([f[0], f[1], f[2], f[3], f[4], f[6], f[9]]))
    summ = json.loads(next(ln for ln in out.splitlines() if ln.startswith("DRIVER_SUMMARY"))[len("DRIVER_SUMMARY "):])
    return iters, summ, out


@pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref not built (needs /root/reference at build time)")
@pytest.mark.parametrize("name,problem,n", [("hs071", "hs071", 0), ("lukvle1_100", "LukVlE1", 100), ("mbndry1_8", "MBndryCntrl1", 8),
                                            ("lukvle1_10000", "LukVlE1", 10000), ("mbndry1_100", "MBndryCntrl1", 100),
                                            ("lukvle1_1000000", "LukVlE1", 1000000),     # the north-star target instance
                                            ("lukvli1_10000", "LukVlI1", 10000), ("lukvle5_10000", "LukVlE5", 10000), ("mbndry2_100", "MBndryCntrl2", 100),
                                            ("mdist1_100", "MDistCntrl1", 100), ("mbndry3d_12", "MBndryCntrl_3D", 12), ("mbndry3d_30", "MBndryCntrl_3D", 30), ("mbndry3d_50", "MBndryCntrl_3D", 50), ("mbndry1_300", "MBndryCntrl1", 300),
                                            ("mbndry1_700", "MBndryCntrl1", 700),       # BASELINE.json configs[4] stand-in (KKT dim 982 800)
                                            ("mbndry3d_100", "MBndryCntrl_3D", 100),    # round 6: KKT dim 2 060 000, fronts up to 22 448 rows, 19.2 TFlop per factorisation; 109 GiB on the device with the contribution blocks recycled (182 without); the reference run behind the golden table took 1 h 35 min
                                            ("mbndry3d_78", "MBndryCntrl_3D", 78)])     # SURVEY 8(d)-5's 3-D instance (solve_problem.cpp:56): KKT dim 985 608, fronts up to 13 598 rows, 4.4 TFlop per factorisation, a 74 GiB factor + contribution-block pool
def test_iteration_sequence_matches_reference_cpu_run(name, problem, n, golden_dir):
    gold = open(os.path.join(golden_dir, name + ".iters")).read().splitlines()
    gsum = json.load(open(os.path.join(golden_dir, name + ".summary")))
    iters, summ, out = run_driver(problem, n)
    assert "EXIT: Optimal Solution Found." in out
    assert summ["iterations"] == gsum["iterations"]
    assert abs(summ["objective"] - gsum["objective"]) <= 1e-8 * max(1.0, abs(gsum["objective"]))
    assert len(iters) == len(gold)
    for a, b in zip(iters, gold):
        fa, fb = a.split(), b.split()
        # iteration number, lg(mu), lg(rg) (the inertia-correction trace) and #line-search steps: identical strings;
        # objective to 1e-7 relative; inf_pr / inf_du to ONE UNIT of the last of the three printed digits (1e-2 relative), with a 1e-11
        # floor for values that sit at rounding level (an infeasibility of 2e-15 is noise of the last solve, not an algorithmic quantity)
        assert (fa[0], fa[4], fa[5], fa[6]) == (fb[0], fb[4], fb[5], fb[6]), f"{a}   |   {b}"
        assert abs(float(fa[1]) - float(fb[1])) <= 1e-7 * max(1.0, abs(float(fb[1]))), f"{a}   |   {b}"
        for k in (2, 3):
            x, y = float(fa[k]), float(fb[k])
            assert abs(x - y) <= 1e-2 * max(x, y) + 1e-11, f"{a}   |   {b}"


# Every problem class the reference's driver registers (examples/ScalableProblems/solve_problem.cpp:28-91) that its own CPU run solves: inequality variants (slacks),
# Neumann boundary control, distributed control, the 27-point 3-D stencils, the parabolic control problem; goldens: tests/golden/make_golden.sh, round 6.
SWEEP = ([(f"lukvl{v}{k}_10000", f"LukVl{v.upper()}{k}", 10000) for k in (2, 6, 7) for v in "ei"] + [(f"lukvl{v}{k}_9998", f"LukVl{v.upper()}{k}", 9998) for k in (3, 4) for v in "ei"]
         + [("lukvli5_10000", "LukVlI5", 10000)] + [(f"mbndry{k}_100", f"MBndryCntrl{k}", 100) for k in (3, 4, 5, 6, 7, 8)]
         + [(f"mdist{k}_100", f"MDistCntrl{k}", 100) for k in ("2", "3", "3a", "4", "5", "6a", "4a", "5a", "6")]
         + [("mbndry3d27_12", "MBndryCntrl_3D_27", 12), ("mbndry3d27bt_12", "MBndryCntrl_3D_27BT", 12), ("mbndry3dsin_12", "MBndryCntrl_3Dsin", 12), ("mpara5_2_3_40", "MPara5_2_3", 40)])


# Five of those classes do NOT print the reference's (MKL PARDISO) table with this backend -- nor would the reference with another of its own linear solvers:
#   LukVlE4 / LukVlI4   the KKT matrix of iteration 1 is numerically singular: oracle/ldlt_oracle.c finds a zero pivot in it (u = 0.01), this backend answers
#                       SYMSOLVER_SINGULAR and Ipopt regularises (lg(rg) = -4.0 in the table's second line); MKL PARDISO perturbs the pivot silently and reports
#                       success (static pivoting, IpPardisoMKLSolverInterface.cpp:555 also hides a wrong inertia) -- the trajectories part there
#   MDistCntrl5 / 5a, MBndryCntrl_3Dsin   line by line the reference's table until the late-barrier (lg(mu) = -8.6) resp. heavily regularised (lg(rg) = 4.4) systems:
#                       from iteration 11-15 on the objective differs in the 7th digit (two direct solvers agree to ~1e-7 on such a system: bench.py's own
#                       parity_vs_gpu line), then a line-search decision, then the count (49-50 / 39-40 / 48-61 against 48 / 39 / 48)
# (tools/table_diff.py prints where a table is left).  They must still arrive where the reference arrives: same exit, objective to 1e-6.
SOLVER_SENSITIVE = {"lukvle4_9998", "lukvli4_9998", "mdist5_100", "mdist5a_100", "mbndry3dsin_12"}


def _driver_run(problem, n, solver):
    env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1")
    out = subprocess.run([DRIVER, problem, str(n), "--solver", solver], capture_output=True, text=True, timeout=900, cwd="/tmp", env=env).stdout
    iters = []
    for ln in out.splitlines():
        f = ln.split()
        if len(f) >= 10 and f[0].rstrip("r").isdigit() and ln.startswith(" "):
            iters.append(" ".join([f[0], f[1], f[2], f[3], f[4], f[6], f[9]]))
    summ = json.loads(next(ln for ln in out.splitlines() if ln.startswith("DRIVER_SUMMARY"))[len("DRIVER_SUMMARY "):])
    return iters, summ, out


@pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref not built (needs /root/reference at build time)")
@pytest.mark.parametrize("solver", ["mi355x", "mi355x-pd"], ids=["B1", "device-route"])
@pytest.mark.parametrize("name,problem,n", [t for t in SWEEP if t[0] not in SOLVER_SENSITIVE], ids=[t[0] for t in SWEEP if t[0] not in SOLVER_SENSITIVE])
def test_every_scalable_problem_class_reproduces_the_reference_iteration_table(name, problem, n, solver, golden_dir):
    gold = open(os.path.join(golden_dir, name + ".iters")).read().splitlines()
    gsum = json.load(open(os.path.join(golden_dir, name + ".summary")))
    iters, summ, out = _driver_run(problem, n, solver)
    assert "EXIT: Optimal Solution Found." in out, out[-1500:]
    assert summ["iterations"] == gsum["iterations"], (summ["iterations"], gsum["iterations"])
    assert abs(summ["objective"] - gsum["objective"]) <= 1e-8 * max(1.0, abs(gsum["objective"]))
    _same_iterations(iters, gold)


@pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref not built")
@pytest.mark.parametrize("solver", ["mi355x", "mi355x-pd"], ids=["B1", "device-route"])
@pytest.mark.parametrize("name,problem,n", [t for t in SWEEP if t[0] in SOLVER_SENSITIVE], ids=[t[0] for t in SWEEP if t[0] in SOLVER_SENSITIVE])
def test_solver_sensitive_problem_classes_reach_the_reference_optimum(name, problem, n, solver, golden_dir):
    gold = open(os.path.join(golden_dir, name + ".iters")).read().splitlines()
    gsum = json.load(open(os.path.join(golden_dir, name + ".summary")))
    iters, summ, out = _driver_run(problem, n, solver)
    assert "EXIT: Optimal Solution Found." in out, out[-1500:]
    assert summ["status"] == gsum["status"] == 0
    assert abs(summ["objective"] - gsum["objective"]) <= 1e-6 * max(1.0, abs(gsum["objective"])), (summ["objective"], gsum["objective"])
    if name.startswith("lukvl"):
        assert iters[0] == gold[0] and iters[1].split()[5] == "-4.0" and gold[1].split()[5] == "-"      # the singular system of iteration 1: reported here, hidden there
    else:
        _same_iterations(iters[:10], gold[:10])      # the reference's table, line by line, for the first ten iterations


HS071 = os.path.join(ROOT, "oracle", "_ref", "hs071_cpp")


@pytest.mark.skipif(not os.path.exists(HS071), reason="oracle/_ref not built")
def test_hsllib_route_stock_ipopt_loads_our_ma97_symbols(tmp_path, golden_dir):
    """Route B2: the reference's UNMODIFIED test binary + `linear_solver ma97` + `hsllib libmi355x_kkt.so`
    (reference IpMa97SolverInterface.cpp:303-315 dlsym()s our seven ma97_*_d exports)."""
    import ipopt_amd
    (tmp_path / "ipopt.opt").write_text(f"linear_solver ma97\nhsllib {ipopt_amd.library_path()}\nma97_scaling none\n")
    out = subprocess.run([HS071], capture_output=True, text=True, timeout=300, cwd=str(tmp_path)).stdout
    assert "EXIT: Optimal Solution Found." in out, out[-2000:]
    iters = [ln.split() for ln in out.splitlines() if ln.startswith(" ") and len(ln.split()) >= 10 and ln.split()[0].isdigit()]
    gold = open(os.path.join(golden_dir, "hs071.iters")).read().splitlines()
    assert len(iters) == len(gold)
    for f, g in zip(iters, gold):
        g = g.split()
        assert f[0] == g[0] and f[4] == g[4] and f[6] == g[5] and f[9] == g[6]      # iter, lg(mu), lg(rg), ls
        assert abs(float(f[1]) - float(g[1])) <= 1e-7 * max(1.0, abs(float(g[1])))


# ---------------------------------------------------------------------------------------------------------------
# the other plug-in routes on real problems (SURVEY 8(b) B1' and B2) and warm_start_same_structure
# ---------------------------------------------------------------------------------------------------------------
PATCHED = os.path.join(ROOT, "oracle", "_ref", "ipopt_patched_driver")
STOCK = os.path.join(ROOT, "oracle", "_ref", "ref_driver")


def _run(binary, args, cwd):
    env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1")
    out = subprocess.run([binary] + args, capture_output=True, text=True, timeout=900, cwd=str(cwd), env=env).stdout
    iters = []
    for ln in out.splitlines():
        f = ln.split()
        if len(f) >= 10 and f[0].rstrip("r").isdigit() and ln.startswith(" "):
            iters.append(" ".join([f[0], f[1], f[2], f[3], f[4], f[6], f[9]]))
    summ = [json.loads(ln[len("DRIVER_SUMMARY "):]) for ln in out.splitlines() if ln.startswith("DRIVER_SUMMARY")]
    return iters, summ, out


def _same_iterations(iters, gold):
    assert len(iters) == len(gold)
    for a, b in zip(iters, gold):
        fa, fb = a.split(), b.split()
        assert (fa[0], fa[4], fa[5], fa[6]) == (fb[0], fb[4], fb[5], fb[6]), f"{a}   |   {b}"
        assert abs(float(fa[1]) - float(fb[1])) <= 1e-7 * max(1.0, abs(float(fb[1]))), f"{a}   |   {b}"


@pytest.mark.skipif(not os.path.exists(PATCHED), reason="oracle/_ref not built")
def test_registered_linear_solver_mi355x_from_ipopt_opt(tmp_path, golden_dir):
    """Route B1': the reference built WITH oracle/patches/linear_solver_mi355x.patch (one factory arm in
    IpAlgBuilder.cpp:427-526 + option registration, IpLinearSolversRegOp.cpp:84-90) selects the backend from ipopt.opt,
    with the registered mi355x_* options, through the reference's OWN AlgorithmBuilder."""
    (tmp_path / "ipopt.opt").write_text("linear_solver mi355x\nmi355x_pivtol 1e-8\nmi355x_ordering nd\n")
    iters, summ, out = _run(PATCHED, ["LukVlE1", "10000", "--solver", "stock", "--optfile", "ipopt.opt"], tmp_path)
    assert "EXIT: Optimal Solution Found." in out, out[-1500:]
    gsum = json.load(open(os.path.join(golden_dir, "lukvle1_10000.summary")))
    assert summ[0]["iterations"] == gsum["iterations"]
    _same_iterations(iters, open(os.path.join(golden_dir, "lukvle1_10000.iters")).read().splitlines())


@pytest.mark.skipif(not os.path.exists(PATCHED), reason="oracle/_ref not built")
@pytest.mark.parametrize("name,problem,n", [("lukvli1_10000", "LukVlI1", 10000), ("mbndry1_100", "MBndryCntrl1", 100)])
def test_registered_linear_solver_mi355x_device_from_ipopt_opt(name, problem, n, tmp_path, golden_dir):
    """Route B1' for the full device route: `linear_solver mi355x-device` in ipopt.opt makes the patched AlgorithmBuilder use
    Mi355xAugSystemSolver (device-side KKT assembly) and put Mi355xPDSystemSolver (device-resident refinement of the 8-block system)
    in front of its PDFullSpaceSolver (IpAlgBuilder.cpp:568-600, :644-664)."""
    (tmp_path / "ipopt.opt").write_text("linear_solver mi355x-device\nprint_timing_statistics yes\n")
    iters, summ, out = _run(PATCHED, [problem, str(n), "--solver", "stock", "--optfile", "ipopt.opt"], tmp_path)
    assert "EXIT: Optimal Solution Found." in out, out[-1500:]
    gsum = json.load(open(os.path.join(golden_dir, name + ".summary")))
    assert summ[0]["iterations"] == gsum["iterations"]
    _same_iterations(iters, open(os.path.join(golden_dir, name + ".iters")).read().splitlines())
    # the reference's own residual computation never ran: its timer stays at zero unless OUR class started it (it does, around the kernels)
    assert summ[0]["LinearSystemStructureConverter"] == 0.0          # no TripletToCSRConverter, no TSymLinearSolver in this route


@pytest.mark.skipif(not os.path.exists(STOCK), reason="oracle/_ref not built")
@pytest.mark.parametrize("name,problem,n", [("mbndry1_100", "MBndryCntrl1", 100), ("lukvle1_10000", "LukVlE1", 10000)])
def test_hsllib_route_on_scalable_problems_with_default_dynamic_scaling(name, problem, n, tmp_path, golden_dir):
    """Route B2 on real problems: the UNPATCHED reference, `linear_solver ma97` + `hsllib libmi355x_kkt.so`, the MA97 adapter's
    DEFAULT `ma97_scaling dynamic` logic (IpMa97SolverInterface.cpp:725-771,824-840) driving our ma97_*_d symbols."""
    import ipopt_amd
    (tmp_path / "ipopt.opt").write_text(f"linear_solver ma97\nhsllib {ipopt_amd.library_path()}\n")
    iters, summ, out = _run(STOCK, [problem, str(n), "--solver", "stock", "--optfile", "ipopt.opt"], tmp_path)
    assert "EXIT: Optimal Solution Found." in out, out[-1500:]
    gsum = json.load(open(os.path.join(golden_dir, name + ".summary")))
    assert summ[0]["iterations"] == gsum["iterations"]
    assert abs(summ[0]["objective"] - gsum["objective"]) <= 1e-8 * max(1.0, abs(gsum["objective"]))
    _same_iterations(iters, open(os.path.join(golden_dir, name + ".iters")).read().splitlines())


@pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref not built")
def test_warm_start_same_structure_keeps_the_symbolic_analysis(tmp_path):
    """InitializeImpl contract (IpSparseSymLinearSolverInterface.hpp:125; pattern IpMumpsSolverInterface.cpp:191-245): a second
    Optimize with warm_start_same_structure=yes must reuse the analysis: same iterates, no symbolic phase."""
    iters, summ, out = _run(DRIVER, ["MBndryCntrl1", "100", "--solver", "mi355x", "--reoptimize"], tmp_path)
    assert len(summ) == 2 and summ[0]["status"] == 0 and summ[1]["status"] == 0
    assert summ[0]["iterations"] == summ[1]["iterations"]
    assert abs(summ[0]["objective"] - summ[1]["objective"]) <= 1e-12 * max(1.0, abs(summ[0]["objective"]))
    assert summ[0]["LinearSystemSymbolicFactorization"] > 0.0
    assert summ[1]["LinearSystemSymbolicFactorization"] == 0.0
    half = len(iters) // 2
    assert iters[:half] == iters[half:]


@pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref not built")
def test_adapter_multigpu_path_with_rccl_world_of_one(tmp_path, golden_dir, monkeypatch):
    """the adapter's multi-GPU plumbing (mi355x_nranks / rank options, unique-id bootstrap, mi355x_kkt_set_comm_rccl, the
    distributed factor/solve behind MultiSolve) on the one-GPU box: an RCCL communicator of size 1"""
    monkeypatch.setenv("MI355X_KKT_FORCE_MULTI", "1")
    monkeypatch.setenv("MI355X_KKT_COMM_FILE", str(tmp_path / "comm_id"))
    iters, summ, out = _run(DRIVER, ["LukVlE1", "10000", "--solver", "mi355x", "--set", "mi355x_nranks", "1", "--set", "mi355x_rank", "0"], tmp_path)
    assert "EXIT: Optimal Solution Found." in out, out[-1500:]
    _same_iterations(iters, open(os.path.join(golden_dir, "lukvle1_10000.iters")).read().splitlines())


@pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref not built")
@pytest.mark.parametrize("name,problem,n", [("hs071", "hs071", 0), ("lukvle1_10000", "LukVlE1", 10000), ("mbndry1_100", "MBndryCntrl1", 100)])
def test_custom_aug_system_solver_with_device_side_assembly(name, problem, n, tmp_path, golden_dir):
    """Route (ii) + SURVEY 8(f)1: Mi355xAugSystemSolver (custom AugSystemSolver, IpAlgBuilder.cpp:82-88,576-584) uploads W / J /
    Sigma pieces only when their tag changed and assembles the KKT values on the GPU.  Same iterates as the reference CPU run;
    hs071's inertia-correction retries (5 trial factorisations in iteration 1) must upload nothing."""
    iters, summ, out = _run(DRIVER, [problem, str(n), "--solver", "mi355x-aug", "--set", "print_level", "5"] +
                            (["--set", "tol", "3.82e-6", "--set", "mu_strategy", "adaptive"] if problem == "hs071" else []), tmp_path)
    assert "EXIT: Optimal Solution Found." in out, out[-1500:]
    gsum = json.load(open(os.path.join(golden_dir, name + ".summary")))
    assert summ[0]["iterations"] == gsum["iterations"]
    assert abs(summ[0]["objective"] - gsum["objective"]) <= 1e-8 * max(1.0, abs(gsum["objective"]))
    _same_iterations(iters, open(os.path.join(golden_dir, name + ".iters")).read().splitlines())
    aug = json.loads(next(ln for ln in out.splitlines() if ln.startswith("AUG_STATS"))[len("AUG_STATS "):])
    if problem == "hs071":
        assert aug["factorizations_without_upload"] >= 4          # the delta_x escalation 1e-4 -> 1e-2 -> 1 -> 100 of iteration 1


@pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref not built")
@pytest.mark.parametrize("name,problem,n", [("hs071", "hs071", 0), ("lukvle1_100", "LukVlE1", 100), ("lukvle1_10000", "LukVlE1", 10000),
                                            ("mbndry1_100", "MBndryCntrl1", 100), ("lukvli1_10000", "LukVlI1", 10000), ("mbndry2_100", "MBndryCntrl2", 100),
                                            ("mdist1_100", "MDistCntrl1", 100), ("lukvle1_1000000", "LukVlE1", 1000000)])
def test_device_resident_primal_dual_solver(name, problem, n, tmp_path, golden_dir):
    """Mi355xPDSystemSolver through the reference's virtual PDSystemSolverFactory (IpAlgBuilder.hpp:138) -- reduce /
    solve / expand / residual / refinement of the 8-block system on the device (mi355x_kkt_pd_*), the reference's own perturbation
    handler and refinement control.  Same iteration table as the reference CPU run (bounds on x: LukVlI1, MBndryCntrl*, hs071;
    inequality constraints with slacks: hs071), every Solve answered on the device."""
    iters, summ, out = _run(DRIVER, [problem, str(n), "--solver", "mi355x-pd", "--set", "print_level", "5"] +
                            (["--set", "tol", "3.82e-6", "--set", "mu_strategy", "adaptive"] if problem == "hs071" else []), tmp_path)
    assert "EXIT: Optimal Solution Found." in out, out[-1500:]
    gsum = json.load(open(os.path.join(golden_dir, name + ".summary")))
    assert summ[0]["iterations"] == gsum["iterations"]
    assert abs(summ[0]["objective"] - gsum["objective"]) <= 1e-8 * max(1.0, abs(gsum["objective"]))
    _same_iterations(iters, open(os.path.join(golden_dir, name + ".iters")).read().splitlines())
    pd = json.loads(next(ln for ln in out.splitlines() if ln.startswith("PD_STATS"))[len("PD_STATS "):])
    assert pd["device_solves"] >= gsum["iterations"] and pd["host_solves"] == 0
    assert pd["refinement_steps"] >= gsum["iterations"]           # min_refinement_steps = 1 per search direction (IpPDFullSpaceSolver.cpp:46-52;
                                                                  # the adaptive-mu oracle of hs071 also solves with allow_inexact: no refinement there)


@pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref not built")
def test_device_route_hands_uncovered_modes_to_the_reference_solver(tmp_path):
    """What Mi355xPDSystemSolver does not cover is answered by the reference PDFullSpaceSolver it wraps (same augmented-system solver, same
    perturbation handler): the inertia-free curvature test (neg_curv_test_tol > 0, IpPDFullSpaceSolver.cpp:592-639) works on host vectors;
    a second optimisation of the same structure (warm_start_same_structure) goes back to the device."""
    iters, summ, out = _run(DRIVER, ["hs071", "0", "--solver", "mi355x-pd", "--set", "neg_curv_test_tol", "1e-12"], tmp_path)
    assert "EXIT: Optimal Solution Found." in out, out[-1500:]
    pd = json.loads(next(ln for ln in out.splitlines() if ln.startswith("PD_STATS"))[len("PD_STATS "):])
    assert pd["device_solves"] == 0 and pd["host_solves"] > 0
    iters, summ, out = _run(DRIVER, ["LukVlI1", "10000", "--solver", "mi355x-pd", "--reoptimize"], tmp_path)
    assert out.count("EXIT: Optimal Solution Found.") == 2, out[-1500:]
    assert summ[0]["iterations"] == summ[1]["iterations"] and summ[1]["LinearSystemSymbolicFactorization"] == 0.0
    pd = json.loads(next(ln for ln in out.splitlines() if ln.startswith("PD_STATS"))[len("PD_STATS "):])
    assert pd["host_solves"] == 0 and pd["device_solves"] >= 2 * summ[0]["iterations"]


@pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref not built")
def test_device_route_reused_for_another_problem_without_warm_start(tmp_path):
    """the AlgorithmBuilder (and with it the cached Mi355xAugSystemSolver) is reused for a second NLP with OTHER bounds and no
    warm_start_same_structure: the new handle must not inherit the first problem's primal-dual workspace request (dimensions, bound
    positions).  LukVlI1 (constraints in [-1, 0]: n_sL = n_sU = m) is followed by the same problem with the upper bound removed
    (n_sU = 0); the second solve must equal a fresh solve of that second problem, every Solve answered on the device."""
    iters, summ, out = _run(DRIVER, ["LukVlI1", "2000", "--solver", "mi355x-pd", "--then-bounds", "-1", "1e20"], tmp_path)
    assert out.count("EXIT: Optimal Solution Found.") == 2, out[-2000:]
    pd = json.loads(next(ln for ln in out.splitlines() if ln.startswith("PD_STATS"))[len("PD_STATS "):])
    assert pd["host_solves"] == 0
    it2, s2, out2 = _run(DRIVER, ["LukVlI1u", "2000", "--solver", "mi355x-pd"], tmp_path)
    assert "EXIT: Optimal Solution Found." in out2
    assert summ[1]["iterations"] == s2[0]["iterations"]
    assert abs(summ[1]["objective"] - s2[0]["objective"]) <= 1e-9 * max(1.0, abs(s2[0]["objective"]))


@pytest.mark.skipif(not os.path.exists(PATCHED), reason="oracle/_ref not built")
def test_dependency_detector_mi355x_removes_the_dependent_constraint(tmp_path):
    """ProvidesDegeneracyDetection / DetermineDependentRows (IpSparseSymLinearSolverInterface.hpp:240-255) through
    the reference's TSymDependencyDetector: on a 6-variable NLP whose third equality constraint is the sum of the first two,
    `dependency_detector mi355x` (patched TNLPAdapter arm) must find exactly one dependent row and Ipopt must then converge."""
    iters, summ, out = _run(PATCHED, ["deptest", "0", "--solver", "stock", "--set", "linear_solver", "mi355x", "--set", "dependency_detector", "mi355x"], tmp_path)
    assert "Detected 1 linearly dependent equality constraints; taking those out." in out, out[-2000:]
    assert "EXIT: Optimal Solution Found." in out
    # min sum (x_i - i)^2 s.t. x1 + x2 = 1, x3 + x4 = 2:  x = (0, 1, 0.5, 1.5, 5, 6), f = 2 + 12.5
    assert abs(summ[0]["objective"] - 14.5) <= 1e-8


@pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref not built")
@pytest.mark.parametrize("on_demand", ["yes", "no"])
def test_outer_scaling_through_the_reference_scaling_hook(on_demand, tmp_path, golden_dir):
    """Mi355xTSymScalingMethod (device Ruiz) plugged into the reference's TSymLinearSolver; with
    linear_scaling_on_demand=yes (default) it stays off unless Ipopt asks for more quality (IpTSymLinearSolver.cpp:429-441),
    with =no every matrix is scaled outside the backend.  The internal equilibration is switched off in both runs."""
    iters, summ, out = _run(DRIVER, ["MBndryCntrl1", "100", "--solver", "mi355x", "--set", "mi355x_outer_scaling", "yes", "--set", "mi355x_scaling", "none",
                                     "--set", "linear_scaling_on_demand", on_demand], tmp_path)
    assert "EXIT: Optimal Solution Found." in out, out[-1500:]
    gsum = json.load(open(os.path.join(golden_dir, "mbndry1_100.summary")))
    assert summ[0]["iterations"] == gsum["iterations"]
    assert abs(summ[0]["objective"] - gsum["objective"]) <= 1e-8 * max(1.0, abs(gsum["objective"]))


@pytest.mark.skipif(not os.path.exists(STOCK), reason="oracle/_ref not built")
@pytest.mark.parametrize("scaling", ["mc64", "mc77"])
def test_hsllib_route_with_explicit_ma97_scaling(scaling, tmp_path, golden_dir):
    """Route B2 with the scaling the MA97 adapter can request explicitly: `ma97_scaling mc64` (control.scaling = 1 -> our
    maximum-product matching scaling, factors written to scale[]) and `mc77` (= Ruiz equilibration on the device)."""
    import ipopt_amd
    (tmp_path / "ipopt.opt").write_text(f"linear_solver ma97\nhsllib {ipopt_amd.library_path()}\nma97_scaling {scaling}\n")
    iters, summ, out = _run(STOCK, ["MBndryCntrl1", "100", "--solver", "stock", "--optfile", "ipopt.opt"], tmp_path)
    assert "EXIT: Optimal Solution Found." in out, out[-1500:]
    gsum = json.load(open(os.path.join(golden_dir, "mbndry1_100.summary")))
    assert summ[0]["iterations"] == gsum["iterations"]
    assert abs(summ[0]["objective"] - gsum["objective"]) <= 1e-8 * max(1.0, abs(gsum["objective"]))


@pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref not built")
def test_matching_scaling_reused_across_an_ipopt_run_at_kkt_dimension_2e5(tmp_path, golden_dir):
    """`mi355x_scaling matching` = maximum-product matching scaling computed at the first factorisation and REUSED (MA97's
    '...-reuse' switches, IpMa97SolverInterface.cpp:725-771) on LukVlE1 n = 10^5 (KKT dimension 2 * 10^5): same iteration count and objective as the
    reference run, and the factorisation timer shows ONE matching, not one per factorisation (the host algorithm costs ~50 ms at this size)."""
    iters, summ, out = _run(DRIVER, ["LukVlE1", "100000", "--solver", "mi355x", "--set", "mi355x_scaling", "matching"], tmp_path)
    assert "EXIT: Optimal Solution Found." in out, out[-1500:]
    _, summ_always, out2 = _run(DRIVER, ["LukVlE1", "100000", "--solver", "mi355x", "--set", "mi355x_scaling", "matching-always"], tmp_path)
    assert "EXIT: Optimal Solution Found." in out2
    assert summ[0]["iterations"] == summ_always[0]["iterations"]
    assert abs(summ[0]["objective"] - summ_always[0]["objective"]) <= 1e-8 * max(1.0, abs(summ_always[0]["objective"]))
    # one matching against one per factorisation
    assert summ[0]["LinearSystemFactorization"] < 0.6 * summ_always[0]["LinearSystemFactorization"], (summ[0], summ_always[0])


@pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref not built")
def test_device_matching_scaling_end_to_end_at_kkt_dimension_2e6(tmp_path, golden_dir):
    """SURVEY 8(f) f3 / VERDICT r03 item 7: the matching scaling ON THE DEVICE through the adapter -- `mi355x_scaling matching-device-always` (scaling mode 5:
    the auction runs at EVERY factorisation) on LukVlE1 n = 10^6 (KKT dimension 2 * 10^6, the 10^6-variable target): the run converges to the golden
    objective in the golden iteration count, and the whole LinearSystemFactorization timer -- seven factorisations, each with its matching -- stays below
    what ONE host matching costs at this size (0.67 s)."""
    iters, summ, out = _run(DRIVER, ["LukVlE1", "1000000", "--solver", "mi355x", "--set", "mi355x_scaling", "matching-device-always"], tmp_path)
    assert "EXIT: Optimal Solution Found." in out, out[-1500:]
    gsum = json.load(open(os.path.join(golden_dir, "lukvle1_1000000.summary")))
    assert summ[0]["iterations"] <= gsum["iterations"] + 2                   # (another scaling: the iterates may differ in their last digits)
    assert abs(summ[0]["objective"] - gsum["objective"]) <= 1e-6 * max(1.0, abs(gsum["objective"]))
    assert summ[0]["LinearSystemFactorization"] < 0.4, summ[0]


@pytest.mark.skipif(not os.path.exists(STOCK), reason="oracle/_ref not built")
def test_hsllib_route_mc64_on_the_device(tmp_path, golden_dir, monkeypatch):
    """Route B2, `ma97_scaling mc64` answered by the device auction (MI355X_KKT_MA97_MATCHING=device): same iteration count and objective as the reference run"""
    import ipopt_amd
    monkeypatch.setenv("MI355X_KKT_MA97_MATCHING", "device")
    (tmp_path / "ipopt.opt").write_text(f"linear_solver ma97\nhsllib {ipopt_amd.library_path()}\nma97_scaling mc64\n")
    iters, summ, out = _run(STOCK, ["MBndryCntrl1", "100", "--solver", "stock", "--optfile", "ipopt.opt"], tmp_path)
    assert "EXIT: Optimal Solution Found." in out, out[-1500:]
    gsum = json.load(open(os.path.join(golden_dir, "mbndry1_100.summary")))
    assert summ[0]["iterations"] == gsum["iterations"]
    assert abs(summ[0]["objective"] - gsum["objective"]) <= 1e-8 * max(1.0, abs(gsum["objective"]))


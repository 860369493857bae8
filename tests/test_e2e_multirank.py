"""-m gpu: BASELINE configs[4] in the form a one-GPU box allows -- SEVERAL Ipopt processes (the unmodified reference host,
oracle/_ref/ipopt_mi355x_driver) share every KKT factorisation: one process per rank, each runs the same deterministic
algorithm, the backend shards the elimination tree over the ranks and sums the Schur contributions at the subtree joins.
The ranks share cuda:0 here, so the adapter's communicator is `mi355x_comm shm` (host-staged sums over POSIX shared memory;
RCCL refuses two ranks on one device) -- the rendez-vous, the rank options, the distributed factor / solve behind MultiSolve
and behind the device routes are exactly those of the 8-GPU run (communicator-in-the-adapter precedent:
reference IpMumpsSolverInterface.cpp:58-75).  EVERY rank must print the reference CPU run's iteration table."""
import json
import os
import subprocess
import uuid

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ipopt_mi355x_driver")


def _table(out):
    iters = []
    for ln in out.splitlines():
        f = ln.split()
        if len(f) >= 10 and f[0].rstrip("r").isdigit() and ln.startswith(" "):
            iters.append(" ".join([f[0], f[1], f[2], f[3], f[4], f[6], f[9]]))
    return iters


def run_ranks(problem, n, solver, world, tmp_path, extra=()):
    env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1", MI355X_KKT_JOB_ID=uuid.uuid4().hex,
               MI355X_KKT_COMM_FILE=str(tmp_path / "comm_id"), MI355X_KKT_SHM_TIMEOUT_S="240")
    procs = []
    for rk in range(world):
        d = tmp_path / f"rank{rk}"
        d.mkdir()
        procs.append(subprocess.Popen([DRIVER, problem, str(n), "--solver", solver, "--set", "mi355x_nranks", str(world), "--set", "mi355x_rank", str(rk),
                                       "--set", "mi355x_comm", "shm", "--set", "mi355x_device", "0", *extra],
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=str(d), env=env))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=900)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return outs


def _same_iterations(iters, gold):
    assert len(iters) == len(gold)
    for a, b in zip(iters, gold):
        fa, fb = a.split(), b.split()
        assert (fa[0], fa[4], fa[5], fa[6]) == (fb[0], fb[4], fb[5], fb[6]), f"{a}   |   {b}"
        assert abs(float(fa[1]) - float(fb[1])) <= 1e-7 * max(1.0, abs(float(fb[1]))), f"{a}   |   {b}"


@pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref not built (needs /root/reference at build time)")
@pytest.mark.parametrize("solver", ["mi355x", "mi355x-aug", "mi355x-pd"], ids=["B1", "device-assembly", "device-route"])
@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("name,problem,n", [("lukvle1_10000", "LukVlE1", 10000), ("mbndry1_100", "MBndryCntrl1", 100)])
def test_every_rank_of_a_multi_rank_ipopt_run_prints_the_reference_iteration_table(name, problem, n, world, solver, tmp_path, golden_dir):
    gold = open(os.path.join(golden_dir, name + ".iters")).read().splitlines()
    gsum = json.load(open(os.path.join(golden_dir, name + ".summary")))
    outs = run_ranks(problem, n, solver, world, tmp_path)
    tables = []
    for rk, out in enumerate(outs):
        assert "EXIT: Optimal Solution Found." in out, f"rank {rk}:\n" + out[-2500:]
        summ = json.loads(next(ln for ln in out.splitlines() if ln.startswith("DRIVER_SUMMARY"))[len("DRIVER_SUMMARY "):])
        assert summ["iterations"] == gsum["iterations"]
        assert abs(summ["objective"] - gsum["objective"]) <= 1e-8 * max(1.0, abs(gsum["objective"]))
        tables.append(_table(out))
        _same_iterations(tables[-1], gold)
        if solver == "mi355x-pd":
            pd = json.loads(next(ln for ln in out.splitlines() if ln.startswith("PD_STATS"))[len("PD_STATS "):])
            assert pd["host_solves"] == 0 and pd["device_solves"] >= gsum["iterations"]
    # the sums are formed in rank order on every rank: the ranks do not merely agree to three digits, they print the same lines
    assert all(t == tables[0] for t in tables)


@pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref not built")
def test_a_rank_that_never_arrives_fails_the_others_instead_of_hanging_them(tmp_path):
    """rank 1 of 2 is never started: rank 0 must come back with an error within the rendez-vous time-out, not hang the box"""
    env = dict(os.environ, MI355X_KKT_JOB_ID=uuid.uuid4().hex, MI355X_KKT_COMM_FILE=str(tmp_path / "comm_id"), MI355X_KKT_SHM_TIMEOUT_S="3")
    out = subprocess.run([DRIVER, "LukVlE1", "100", "--solver", "mi355x", "--set", "mi355x_nranks", "2", "--set", "mi355x_rank", "0", "--set", "mi355x_comm", "shm",
                          "--set", "mi355x_device", "0"], capture_output=True, text=True, timeout=120, cwd=str(tmp_path), env=env)
    text = out.stdout + out.stderr
    assert "EXIT: Optimal Solution Found." not in text
    assert "not every rank attached" in text or "set_comm_shm failed" in text, text[-1500:]

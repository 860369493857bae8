"""-m gpu: the HIP multi-GPU code path (factor_local / arena / factor_top / sharded solves) through the C ABI.
The GPU test box has ONE MI355X, so the ranks share cuda:0 and the collectives go over gloo (RCCL refuses two ranks
on one device); kernels, arena layout and the collective sequence are exactly those of the 8-GPU run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.support import kktgen

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, case, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ipopt_amd.multigpu import DistributedKKT, HipEngine
    n, r, c, v, neg = case()
    K = kktgen.to_scipy(n, r, c, v)
    eng = HipEngine(rank, world, 0)
    eng.analyse(n, r, c, v)
    D = DistributedKKT(eng, dist)
    dv = torch.tensor(v, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    out = []
    for rep in range(2):
        st, nneg = D.factor(dv)
        xt = np.random.default_rng(rep).standard_normal(n)
        b = K @ xt
        db = torch.tensor(b, dtype=torch.float64, device="cuda"); torch.cuda.synchronize()
        D.solve(db)
        x = db.cpu().numpy()
        res = float(np.abs(K @ x - b).max() / (abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max()))
        out.append((st, nneg, res, float(np.abs(x - xt).max())))
    if rank == 0:
        # single-GPU factorisation of the same system for comparison
        import ipopt_amd
        s1 = ipopt_amd.KKTSolver(device=0); s1.initialize_structure(n, r, c, vals=v); s1.values()[:] = v
        x1 = (K @ np.ones(n)).copy(); st1 = s1.multi_solve(True, x1, True, neg)
        ret.put((out, neg, st1, s1.number_of_neg_evals(), int(eng.arena().numel())))
    dist.barrier()
    dist.destroy_process_group()


def _case_grid():
    return kktgen.grid_kkt(40, 36, dof=2, ncon=1, seed=8)      # top fronts go through the blocked (big) path


def _case_band():
    return kktgen.lukvl_like(20000, seed=9)


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("case", [_case_grid, _case_band], ids=["grid", "band"])
def test_hip_multigpu_path_matches_single_gpu(world, case):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(rk, world, port, case, ret)) for rk in range(world)]
    for p in procs:
        p.start()
    out, neg, st1, neg1, arena = ret.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert st1 == 0 and neg1 == neg and arena > 0
    for st, nneg, res, err in out:
        assert st == 0 and nneg == neg       # inertia: exact, summed over ranks
        assert res <= 1e-12                  # same tolerance as the single-GPU path

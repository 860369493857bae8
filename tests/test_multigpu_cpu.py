"""-m 'not gpu': the N > 1 path on CPU -- world_size 2, 4 and 8 (the driver's scaling run) over gloo.  The product's collective sequence
(ipopt_amd.multigpu.DistributedKKT: all-reduce of the top arena at the subtree joins, replicated top, mirrored solve)
runs unchanged; the per-rank numeric engine is the numpy walk of the same symbolic structures
(tests/support/mirror_mg.py), because there is no GPU here.  The HIP engine is exercised by test_multigpu_gpu.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.support import kktgen


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, case, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ipopt_amd.multigpu import DistributedKKT
    from tests.support.mirror_mg import MirrorEngine
    n, r, c, v, neg = case()
    K = kktgen.to_scipy(n, r, c, v)
    eng = MirrorEngine(rank, world)
    eng.analyse(n, r, c, v)
    D = DistributedKKT(eng, dist)
    st, nneg = D.factor(v)
    xs = []
    for seed in (0, 1):
        xt = np.random.default_rng(seed).standard_normal(n)
        b = torch.from_numpy(K @ xt)
        D.solve(b)
        xs.append(float(np.abs(b.numpy() - xt).max()))
    own = eng.sym["owner"]
    if rank == 0:
        ret.put((st, nneg, neg, xs, int((own < 0).sum()), sorted(set(own.tolist()))))
    dist.barrier()
    dist.destroy_process_group()


def _case_grid():
    return kktgen.grid_kkt(20, 18, dof=2, ncon=1, seed=5)


def _case_band():
    return kktgen.lukvl_like(1500, seed=6)


@pytest.mark.parametrize("world,case", [(2, _case_grid), (2, _case_band), (4, _case_grid), (4, _case_band), (8, _case_grid)],
                         ids=["2-grid", "2-band", "4-grid", "4-band", "8-grid"])
def test_subtree_sharded_factor_solve_over_gloo(world, case):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(rk, world, port, case, ret)) for rk in range(world)]
    for p in procs:
        p.start()
    st, nneg, neg, errs, ntop, owners = ret.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert st == 0 and nneg == neg
    assert max(errs) <= 1e-8
    assert ntop >= 1 and owners[0] == -1 and owners[1:] == list(range(world))

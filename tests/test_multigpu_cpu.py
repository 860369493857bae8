"""-m 'not gpu': the N > 1 path on CPU -- world_size 2, 3, 4 and 8 (the driver's scaling run) over gloo.  The product's collective sequence
(ipopt_amd.multigpu.DistributedKKT: per exchange step the all-reduce of the arena squares at the joins, the replicated fronts, mirrored
solve) runs unchanged, with the classic mapping (one top replicated on all ranks: one step) and with the subtree-to-subcube mapping
(option subcube: a top front is held by the ranks beneath it only; one step per bisection of the machine); the per-rank numeric engine
is the numpy walk of the same symbolic structures (tests/support/mirror_mg.py), because there is no GPU here.  The HIP engine is
exercised by test_multigpu_gpu.py."""
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


def _worker(rank, world, port, case, subcube, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ipopt_amd.multigpu import DistributedKKT
    from tests.support.mirror_mg import MirrorEngine
    n, r, c, v, neg = case()
    K = kktgen.to_scipy(n, r, c, v)
    eng = MirrorEngine(rank, world, subcube=subcube)
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
    held = [int(sum(1 for s in range(eng.nsn) if eng.held(s)))]
    allheld = [None] * world
    dist.all_gather_object(allheld, held[0])
    if rank == 0:
        ret.put((st, nneg, neg, xs, int((own < 0).sum()), sorted(set(own.tolist())), eng.num_steps(), allheld))
    dist.barrier()
    dist.destroy_process_group()


def _case_grid():
    return kktgen.grid_kkt(20, 18, dof=2, ncon=1, seed=5)


def _case_band():
    return kktgen.lukvl_like(1500, seed=6)


@pytest.mark.parametrize("world,case,subcube", [(2, _case_grid, 0), (2, _case_band, 0), (4, _case_grid, 0), (4, _case_band, 0), (8, _case_grid, 0),
                                                (4, _case_grid, 1), (4, _case_band, 1), (3, _case_grid, 1), (8, _case_grid, 1), (8, _case_band, 1)],
                         ids=["2-grid", "2-band", "4-grid", "4-band", "8-grid", "4-grid-subcube", "4-band-subcube", "3-grid-subcube", "8-grid-subcube", "8-band-subcube"])
def test_subtree_sharded_factor_solve_over_gloo(world, case, subcube):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(rk, world, port, case, subcube, ret)) for rk in range(world)]
    for p in procs:
        p.start()
    st, nneg, neg, errs, ntop, owners, nsteps, held = ret.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert st == 0 and nneg == neg
    assert max(errs) <= 1e-8
    assert ntop >= 1 and owners[0] == -1 and owners[1:] == list(range(world))
    if subcube and world >= 4:
        assert nsteps >= 2                      # ranges of ranks below the whole machine exist ...
        assert min(held) < ntop                 # ... and not every rank holds every replicated front any more
    else:
        assert nsteps == 1 or subcube
        if not subcube:
            assert held == [ntop] * world


def test_both_mappings_on_random_systems_at_two_to_nine_ranks():
    """tools/fuzz_subcube.py, short: every rank of a world is a numpy engine in THIS process and the all-reduces are sums over their buffers
    -- random grid / band systems, 2..9 ranks (odd counts, ranges that skip a bisection level), classic and subtree-to-subcube mapping: inertia
    summed over the ranks and the solution on every rank"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_subcube", os.path.join(root, "tools", "fuzz_subcube.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    mod.main(cases=8, seed=3)

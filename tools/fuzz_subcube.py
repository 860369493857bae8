"""Randomised check of the multi-GPU mappings on CPU, no process group: all ranks of a world are numpy engines in ONE process, the
all-reduces are sums over their buffers.  Random grid / band systems, 2..9 ranks, classic and subtree-to-subcube mapping; checks the
inertia and the solution of every rank.  usage: python tools/fuzz_subcube.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests.support import kktgen
from tests.support.mirror_mg import MirrorEngine


def run_world(world, subcube, n, r, c, v, neg, rng):
    K = kktgen.to_scipy(n, r, c, v)
    engs = [MirrorEngine(rk, world, subcube=subcube) for rk in range(world)]
    for e in engs:
        e.analyse(n, r, c, v)
    steps = engs[0].num_steps()
    assert all(e.num_steps() == steps for e in engs)

    def allreduce(bufs):
        tot = sum(b.numpy().copy() for b in bufs)
        for b in bufs:
            b.numpy()[:] = tot

    def exchange(d, what):
        # range-local: a range of ranks sums its part of the step among ITS ranks; the others hold zeros there and stay out of it
        segs = {}
        for e in engs:
            for lo, g, t in e.exchange_segments(d, what):
                segs.setdefault((lo, g), []).append((e.rank, t))
        for (lo, g), members in segs.items():
            assert sorted(rk for rk, _ in members) == list(range(lo, lo + g)), (lo, g, [rk for rk, _ in members])
            allreduce([t for _, t in members])

    for e in engs:
        e.factor_local(v)
    for d in reversed(range(steps)):
        exchange(d, 0)
        for e in engs:
            e.factor_step(d)
    nneg = sum(e.counters()[0] for e in engs)
    assert nneg == neg, (world, subcube, nneg, neg)
    xt = rng.standard_normal(n)
    rhs = [torch.from_numpy((K @ xt).copy()) for _ in engs]
    for e, b in zip(engs, rhs):
        e.fwd_local(b)
    for d in reversed(range(steps)):
        exchange(d, 1)
        for e in engs:
            e.fwd_step(d)
    for e, b in zip(engs, rhs):
        e.bwd(b)
    allreduce(rhs)
    err = max(float(np.abs(b.numpy() - xt).max()) for b in rhs)
    assert err <= 1e-7 * max(1.0, np.abs(xt).max()), (world, subcube, err)
    return steps


def main(cases=40, seed=0):
    rng = np.random.default_rng(seed)
    for it in range(cases):
        if rng.random() < 0.6:
            nx, ny = int(rng.integers(6, 26)), int(rng.integers(5, 22))
            dof = int(rng.integers(1, 4))
            n, r, c, v, neg = kktgen.grid_kkt(nx, ny, dof=dof, ncon=int(rng.integers(1, dof + 1)), seed=int(rng.integers(1 << 30)))
            what = f"grid {nx}x{ny}"
        else:
            nn = int(rng.integers(50, 1500))
            n, r, c, v, neg = kktgen.lukvl_like(nn, seed=int(rng.integers(1 << 30)))
            what = f"band {nn}"
        world = int(rng.integers(2, 10))
        for sub in (0, 1):
            steps = run_world(world, sub, n, r, c, v, neg, rng)
        print(f"{it:3d} {what:14s} n={n:6d} world={world} subcube steps={steps}: ok", flush=True)
    print("fuzz done")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0)

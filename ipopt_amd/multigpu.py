"""Multi-GPU driver: one process per GPU, elimination-tree subtrees sharded over the ranks, the fronts above them replicated --
on every rank (classic mapping: ONE exchange step) or only on the range of ranks beneath each front (subtree-to-subcube mapping, option
`subcube`: one exchange step per bisection of the machine) -- and the Schur contributions that cross a range boundary combined by one
all-reduce per step (RCCL over xGMI; `torch.distributed` backend "nccl" is RCCL on ROCm).  See DESIGN.md (e).

The collective SEQUENCE is stated here (DistributedKKT: host logic, covered by world_size 2-8 gloo tests on CPU with a numpy engine from
tests/support) and implemented inside the C library behind the ordinary entry points once a communicator is set (CommKKT: what bench.py
--gpus N and the Ipopt adapter use).  torch is plumbing only: device tensors, streams, process group.

    factor:  engine.factor_local(vals)                      # own subtrees + what they contribute to the replicated fronts above them
             for step in deepest ... 0:  all_reduce(arena[step], SUM);  engine.factor_step(step)
             all_reduce(counters)
    solve :  engine.fwd_local(rhs);  for step in deepest ... 0:  all_reduce(top_rhs[step]);  engine.fwd_step(step)
             engine.bwd(rhs);  all_reduce(rhs)
"""
from __future__ import annotations

import ctypes as C
import json
import os
import time

import numpy as np

from . import kkt as _kkt

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _DevArray:
    """zero-copy view of library-owned device memory for torch (CUDA array interface v2)."""

    def __init__(self, ptr: int, n: int, typestr: str = "<f8"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2, "strides": None}


class HipEngine:
    """per-rank numeric engine = one mi355x_kkt handle created with (nranks, rank)."""

    def __init__(self, rank: int, nranks: int, device: int, **opts):
        import torch
        self.torch = torch
        self.s = _kkt.KKTSolver(device=device, nranks=nranks, rank=rank, **opts)
        self.rank, self.nranks = rank, nranks

    def analyse(self, n, row, col, vals):
        self.s.initialize_structure(n, row, col, vals=vals)
        self.n = n

    def _view(self, fn):
        p, nd = C.c_void_p(), C.c_int64()
        if fn(self.s._h, C.byref(p), C.byref(nd)) != 0:
            raise _kkt.KKTError(self.s.last_error())
        if nd.value == 0:
            return self.torch.zeros(0, dtype=self.torch.float64, device="cuda")
        return self.torch.as_tensor(_DevArray(p.value, nd.value), device="cuda")

    def num_steps(self):
        return 1                                         # the phase entry points serve the classic mapping (one replicated top)

    def factor_local(self, dvals):
        if self.s.lib.mi355x_kkt_factor_local(self.s._h, C.c_void_p(dvals.data_ptr())) != 0:
            raise _kkt.KKTError("factor_local: " + self.s.last_error())

    def arena(self, d=0):
        return self._view(self.s.lib.mi355x_kkt_top_arena)

    def exchange_segments(self, d, what):
        """[(rank_lo, nranks_in_range, tensor)] this rank takes part in at step d (what 0: arena squares, 1: top right-hand sides);
        the phase entry points serve the classic mapping: one range, the whole machine"""
        t = self.arena(d) if what == 0 else self.top_rhs(d)
        return [(0, self.nranks, t)] if t.numel() > 0 else []

    def factor_step(self, d):
        neg, zero = C.c_int(0), C.c_int(0)
        if self.s.lib.mi355x_kkt_factor_top(self.s._h, C.byref(neg), C.byref(zero)) != 0:
            raise _kkt.KKTError("factor_top: " + self.s.last_error())
        self._cnt = (neg.value, zero.value)

    def counters(self):
        return self._cnt

    def fwd_local(self, drhs):
        if self.s.lib.mi355x_kkt_solve_fwd_local(self.s._h, C.c_void_p(drhs.data_ptr())) != 0:
            raise _kkt.KKTError("solve_fwd_local: " + self.s.last_error())

    def top_rhs(self, d=0):
        return self._view(self.s.lib.mi355x_kkt_top_rhs)

    def fwd_step(self, d):
        pass                                             # (solve_top_and_bwd runs the replicated forward sweep too)

    def bwd(self, drhs):
        if self.s.lib.mi355x_kkt_solve_top_and_bwd(self.s._h, C.c_void_p(drhs.data_ptr())) != 0:
            raise _kkt.KKTError("solve_top_and_bwd: " + self.s.last_error())

    def counters_tensor(self, neg, zero):
        return self.torch.tensor([neg, zero], dtype=self.torch.int64, device="cuda")

    def sync(self):
        self.torch.cuda.synchronize()


class DistributedKKT:
    """Collective sequence around an engine (HipEngine on GPUs; a numpy engine in the CPU tests): own subtrees, then per exchange step --
    deepest ranges of ranks first -- all-reduce of that step's arena squares and the step's replicated fronts.  Every rank takes part in
    every collective in the same order (ranges it is not in contribute zeros).  The classic replicated top is ONE step; the
    subtree-to-subcube mapping (option subcube) has one per bisection of the machine.  Mirrors numeric.hip: factor_dist / solve_dist."""

    def __init__(self, engine, dist):
        self.e, self.dist = engine, dist
        self.groups = range_groups(dist, engine.nranks)  # (collective: every rank creates every group, in the same order)

    def _exchange(self, d, what):
        # a range of ranks sums its part of the step among itself (its sub-communicator); ranks outside it are not involved
        for lo, g, t in self.e.exchange_segments(d, what):
            if t.numel() > 0:
                self.dist.all_reduce(t, group=self.groups[(lo, g)])
        self.e.sync()

    def factor(self, vals):
        e, dist = self.e, self.dist
        e.factor_local(vals)                             # own subtrees + what they contribute to the replicated fronts above them
        for d in reversed(range(e.num_steps())):
            self._exchange(d, 0)                         # sum of the contributions from outside each front's range of ranks
            e.factor_step(d)
        neg, zero = e.counters()
        cnt = e.counters_tensor(neg, zero)
        dist.all_reduce(cnt)
        e.sync()
        self.num_neg, self.num_zero = int(cnt[0]), int(cnt[1])
        return (1 if self.num_zero > 0 else 0), self.num_neg

    def solve(self, rhs):
        """rhs: full right-hand side, identical on every rank; overwritten by the full solution on every rank."""
        e, dist = self.e, self.dist
        e.fwd_local(rhs)
        for d in reversed(range(e.num_steps())):
            self._exchange(d, 1)
            e.fwd_step(d)
        e.bwd(rhs)                                       # replicated fronts (nothing to exchange on the way down), own subtrees, own solution pieces
        dist.all_reduce(rhs)
        e.sync()
        return rhs


def range_groups(dist, world):
    """{(rank_lo, nranks): process group} for every contiguous range of ranks (the whole world: None = the default group).  new_group is a
    collective over the WORLD, so every rank creates every group in the same order -- which ranges a structure uses is not known to the ranks
    outside them, and a delayed-pivot edit may change them."""
    groups = {(0, world): None}
    for g in range(2, world):
        for lo in range(0, world - g + 1):
            groups[(lo, g)] = dist.new_group(list(range(lo, lo + g)))
    return groups


def gloo_allreduce_callback(torch, dist, world=None):
    """the collectives the C library needs (mi355x_kkt_allreduce_fn and, when `world` is given, mi355x_kkt_allreduce_range_fn), supplied from
    Python over any process group -- used where RCCL cannot run: several ranks sharing one GPU (the single-GPU test box)."""
    def fn(dptr, count, dtype, stream):
        t = torch.as_tensor(_DevArray(dptr, count, "<f8" if dtype == 0 else "<i4"), device="cuda")
        torch.cuda.synchronize()          # the library's stream has produced the buffer
        dist.all_reduce(t)
        torch.cuda.synchronize()
    if world is None:
        return fn
    groups = range_groups(dist, world)

    def fn_range(dptr, count, dtype, stream, lo, g):
        t = torch.as_tensor(_DevArray(dptr, count, "<f8" if dtype == 0 else "<i4"), device="cuda")
        torch.cuda.synchronize()
        dist.all_reduce(t, group=groups[(lo, g)])
        torch.cuda.synchronize()
    return fn, fn_range


class CommKKT:
    """A multi-GPU handle whose collectives run INSIDE the C library (RCCL over xGMI, mi355x_kkt_set_comm_rccl):
    after construction it is used exactly like a single-GPU KKTSolver (factor_device / solve_device2 / multi_solve).
    torch.distributed is only the bootstrap that carries the 128-byte ncclUniqueId from rank 0 to the other ranks."""

    def __init__(self, rank, nranks, device, n, row, col, vals, dist, use_rccl=True, use_shm=False, **opts):
        import torch
        self.s = _kkt.KKTSolver(device=device, nranks=nranks, rank=rank, **opts)
        self.s.initialize_structure(n, row, col, vals=vals)
        if use_shm:
            # the library's own host-staged communicator over POSIX shared memory (ranks of one node, which may share a device)
            box = [_kkt.KKTSolver.comm_shm_id(nranks) if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            self.s.set_comm_shm(box[0])
        elif use_rccl:
            box = [_kkt.KKTSolver.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            self.s.set_comm_rccl(box[0])
        else:
            fn, fn_range = gloo_allreduce_callback(torch, dist, nranks)
            self.s.set_comm_callback(fn, fn_range if "subcomm" not in os.environ.get("MI355X_KKT_DISABLE", "").split(",") else None)


def play_comm_plans(plans):
    """plans[rank] = KKTSolver.comm_plan(rank): play the ranks' collective lists against each other -- a collective completes when EVERY member of its
    communicator has it at the head of its list with the same (what, step, count, dtype).  Returns (collectives completed, {(step, colour): members}) or
    raises AssertionError on a mismatch / when nothing can complete although a list is not empty (= a deadlock).  The property that makes an N-rank run
    of factor_dist / solve_dist hang-free, checkable on a machine without a GPU (tests/test_comm_plan.py; bench.py --gpus N dry run)."""
    SPLIT, WHOLE, NOCOLOR = 0, -2, -1
    world = len(plans)
    subcomm = {}
    for rk in range(world):
        for what, d, colour, gsz, cnt, dt in plans[rk]:
            if what == SPLIT and colour != NOCOLOR:
                subcomm.setdefault((int(d), int(colour)), []).append(rk)
    queues = [[tuple(int(x) for x in rec) for rec in plans[rk] if rec[0] != SPLIT] for rk in range(world)]
    done = 0
    while any(queues):
        progress = False
        for rk in range(world):
            if not queues[rk]:
                continue
            what, d, colour, gsz, cnt, dt = queues[rk][0]
            members = list(range(world)) if colour == WHOLE else subcomm[(d, colour)]
            assert rk in members, (rk, what, d, colour)
            heads = [queues[m][0] if queues[m] else None for m in members]
            if all(h is not None and h[2] == colour and h[1] == d for h in heads):
                assert len(set(heads)) == 1, f"members of communicator (step {d}, colour {colour}) disagree: {set(heads)}"
                for m in members:
                    queues[m].pop(0)
                done += 1
                progress = True
        assert progress, f"deadlock: heads {[q[0] if q else None for q in queues]}"
    return done, subcomm


def bench_dry_run(args, rank, world):
    """bench.py --gpus N with MI355X_KKT_BENCH_DRYRUN=1: everything of the N-rank launch that needs no device -- the launcher, the rendezvous (gloo), the
    analysis with (nranks, rank) on every rank, the collective plan of every rank gathered and played against each other -- and a JSON line that says so.
    What a CPU-only box can prove about first contact with N GPUs (tests/test_bench_tools.py runs it with two ranks)."""
    import torch.distributed as dist
    import bench as B
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wl = "grid_1e5" if args.workload == "auto" else args.workload
    n, r, c, v, neg = B.make_workload(wl)
    subcube = int(os.environ.get("MI355X_KKT_SUBCUBE", "1" if world > 2 else "0"))
    s = _kkt.KKTSolver(device=-1, nranks=world, rank=rank, subcube=subcube)
    s.initialize_structure(n, r, c, vals=v)
    range_local = "subcomm" not in os.environ.get("MI355X_KKT_DISABLE", "").split(",")
    mine = s.comm_plan(rank, range_local).tolist()
    box = [None] * world
    dist.all_gather_object(box, mine)
    I = s.info()
    if rank == 0:
        done, subcomm = play_comm_plans([np.array(p, dtype=np.int64).reshape(-1, 6) for p in box])
        print(json.dumps({"dry_run": True, "what": "launcher + rendezvous + per-rank analysis + collective plans of all ranks played against each other; no HIP call was made",
                          "n_gpus": world, "ranks_seen": len(box), "workload": wl, "kkt_dim": n, "subcube": subcube, "range_local_collectives": range_local,
                          "collectives_per_factor_plus_solve": done, "sub_communicators": [[int(d), members] for (d, col), members in sorted(subcomm.items())],
                          "supernodes": I.num_sn, "plan_ok": True}))
    dist.barrier()
    dist.destroy_process_group()


def partition_model(s, I, own, world):
    """What the partition itself allows (work only, no latency, no communication): every rank does the flops of its own subtrees and of the
    replicated fronts it holds (all of them with the classic mapping, those above its subtrees with the subtree-to-subcube mapping);
    predicted_speedup_bound = total / most loaded rank.  The measured speed-up of the line can be compared with it: the gap is latency of
    the replicated separator chains + the all-reduces."""
    colptr = s.symbolic(1, I.num_sn + 1).astype(np.int64); rowptr = s.symbolic(2, I.num_sn + 1).astype(np.int64)
    glo, gsz, gd = s.symbolic(18, I.num_sn), s.symbolic(19, I.num_sn), s.symbolic(20, I.num_sn)
    k = np.diff(colptr); m = np.diff(rowptr)
    # sum_{j<k} (c_j - 1)(c_j + 2), c_j = m - j
    j = np.arange(int(k.max()) + 1)
    f = np.array([(((mm - j[:kk]) - 1) * ((mm - j[:kk]) + 2)).sum() for kk, mm in zip(k, m)], dtype=np.float64)
    top = own < 0
    held = [top & (glo <= r) & (r < glo + gsz) for r in range(world)]
    per_rank = [float(f[own == r].sum() + f[held[r]].sum()) for r in range(world)]
    total = float(f.sum())
    crit = max(per_rank) if per_rank else total
    steps = int(gd[top].max()) + 1 if top.any() else 1
    # what the exchange steps move.  A join front (one with a child from outside its range) receives the lower triangle of its square, every
    # replicated front its top right-hand side; a range of g ranks sums its part among itself (ring all-reduce: every rank sends and receives
    # 2 (g - 1) / g of the bytes); the steps run one after the other, the ranges of a step side by side.
    par = s.symbolic(4, I.num_sn)
    same = lambda a, b: own[a] < 0 and own[b] < 0 and glo[a] == glo[b] and gsz[a] == gsz[b]
    join = np.zeros(I.num_sn, dtype=bool)
    for ch in range(I.num_sn):
        p = par[ch]
        if p >= 0 and own[p] < 0 and not same(ch, p):
            join[p] = True
    comm = []
    for d in range(steps):
        ranges = sorted(set((int(glo[q]), int(gsz[q])) for q in np.nonzero(top & (gd == d))[0]))
        rows = []
        for lo, g in ranges:
            sel = top & (gd == d) & (glo == lo) & (gsz == g)
            ab = int((m[sel & join] * (m[sel & join] + 1) // 2).sum() * 8); tb = int(m[sel].sum() * 8)
            rows.append({"ranks": [lo, lo + g], "arena_bytes": ab, "rhs_bytes": tb, "bytes_on_every_link_of_the_ring": 2.0 * (g - 1) / g * ab})
        comm.append(rows)
    arena_total = sum(r["arena_bytes"] for rows in comm for r in rows)
    per_link = sum(max((r["bytes_on_every_link_of_the_ring"] for r in rows), default=0.0) for rows in comm)        # critical path: the slowest range of every step
    whole = 2.0 * (world - 1) / world * arena_total                                                             # ... and with ONE all-reduce over the whole machine per step
    return {"exchange": {"arena_bytes_total": arena_total, "arena_bytes_if_full_squares": int(sum((m[top & join] ** 2).sum() for _ in [0]) * 8),
                         "per_step": comm, "ring_bytes_on_the_critical_path": per_link, "ring_bytes_without_sub_communicators": whole,
                         "predicted_ms_per_factorisation": {"at_100_GB/s": per_link / 100e9 * 1e3, "at_200_GB/s": per_link / 200e9 * 1e3},
                         "what": "lower triangles of the join fronts' squares, summed inside the range of ranks that holds each front (one sub-communicator per step)"},
            "flops_total": total, "flops_replicated_fronts": float(f[top].sum()), "flops_replicated_on_the_most_loaded_rank": float(max(f[h].sum() for h in held)) if held else 0.0,
            "flops_most_loaded_rank": crit, "flops_least_loaded_rank": min(per_rank) if per_rank else total,
            "replicated_fronts": int(top.sum()), "replicated_fronts_held_per_rank": [int(h.sum()) for h in held], "exchange_steps": steps,
            "predicted_speedup_bound": total / crit if crit else 1.0,
            "what": "work-only bound of the partition (own subtrees + the replicated fronts a rank holds); latency of the replicated separator chains and the all-reduces come on top"}


def bench_main(args, rank, world, local):
    """bench.py --gpus N (N > 1): launched by torch.distributed.run, one rank per GPU."""
    import torch
    import torch.distributed as dist
    import bench as B
    from tests.support import kktgen

    if os.environ.get("MI355X_KKT_BENCH_DRYRUN"):
        return bench_dry_run(args, rank, world)
    # MI355X_KKT_BENCH_SHARED=1: all ranks on device 0 (a one-GPU box) -- gloo bootstrap, the library's shared-memory communicator instead of RCCL
    # (which refuses two ranks on one device); kernels, partition and collective sequence are those of the N-GPU run, the timings are not
    shared = bool(os.environ.get("MI355X_KKT_BENCH_SHARED"))
    if shared:
        local = 0
    torch.cuda.set_device(local)
    dist.init_process_group("gloo" if shared else "nccl", rank=rank, world_size=world)
    wl = "synth_1e6" if args.workload == "auto" else args.workload
    n, r, c, v, neg = B.make_workload(wl)
    K = kktgen.to_scipy(n, r, c, v)
    b = K @ np.ones(n)
    # the collectives (all-reduce of the top arena / top right-hand sides / solution, RCCL over xGMI) run inside the C library
    subcube = int(os.environ.get("MI355X_KKT_SUBCUBE", "1" if world > 2 else "0"))      # (two ranks: the two mappings coincide)
    ck = CommKKT(rank, world, local, n, r, c, v, dist, use_rccl=not shared, use_shm=shared, subcube=subcube)
    s = ck.s
    I = s.info()
    dv = torch.tensor(v, dtype=torch.float64, device="cuda")
    db = torch.tensor(b, dtype=torch.float64, device="cuda")
    dx = torch.empty_like(db)
    torch.cuda.synchronize()
    NSOLVE = 2

    def step():
        st, nneg, _ = s.factor_device(dv.data_ptr())
        for _ in range(NSOLVE):
            s.solve_device2(db.data_ptr(), dx.data_ptr())
        return st, nneg

    for _ in range(max(args.warmup, 1)):
        st, nneg = step()
    assert st == 0 and nneg == neg, f"rank {rank}: inertia {nneg} != {neg} (status {st})"
    x = dx.cpu().numpy()
    res = float(np.abs(K @ x - b).max() / (abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max()))
    assert res <= 1e-12, f"rank {rank}: scaled residual {res}"
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    dist.barrier(); torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cpu" if shared else "cuda")
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    dt = float(el[0]) / args.steps
    own = s.symbolic(11, I.num_sn)
    flops_step = I.flops_factor + NSOLVE * I.flops_solve
    cinfo = s.comm_info()      # what the communicator itself says: ncclCommCount, whether ncclCommSplit gave every range its sub-communicator
    line = None
    if rank == 0:
        # the same workload on ONE GPU of this node, so that the line carries its own strong-scaling reference
        import ipopt_amd
        os.environ.pop("MI355X_KKT_FORCE_MULTI", None)      # (the 1-rank smoke run of this path forces it; the comparison handle is a plain single-GPU one)
        s1 = ipopt_amd.KKTSolver(device=local)
        s1.initialize_structure(n, r, c, vals=v)
        for _ in range(2):
            s1.factor_device(dv.data_ptr())
            for _ in range(NSOLVE):
                s1.solve_device2(db.data_ptr(), dx.data_ptr())
        torch.cuda.synchronize(); t1 = time.perf_counter()
        reps = max(2, min(args.steps, 5))
        for _ in range(reps):
            s1.factor_device(dv.data_ptr())
            for _ in range(NSOLVE):
                s1.solve_device2(db.data_ptr(), dx.data_ptr())
        torch.cuda.synchronize(); dt1 = (time.perf_counter() - t1) / reps
        prof = s1.profile(3)
        work = B.per_kind_work(s1)
        per_rep = {kn: ms / 3 for kn, (ms, ln) in prof.items() if ln > 0}
        dom = max(per_rep, key=per_rep.get)
        w = work.get(dom, dict(bytes=0, flops=0))
        if dom == "big_schur":
            roof = dict(B.schur_roofline(w, per_rep[dom]), traffic=None)      # (both roofs; `bound` = the lower ceiling at the measured intensity, as on the one-GPU line)
        else:
            ach = w["bytes"] / (per_rep[dom] * 1e-3) / 1e9
            roof = dict(bound="hbm", kernel=dom, achieved=ach, peak=B.HBM_PEAK_GBS, unit="GB/s", frac=ach / B.HBM_PEAK_GBS, traffic=None)
        roof["measured_on"] = "rank 0, single-GPU pass over the same workload (same kernels)"
        line = {
            "metric": "KKT factor+solve GFLOP/s (1 numeric LDL^T factorisation + 2 solves per Ipopt iteration)",
            "value": flops_step / dt / 1e9, "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl, "kkt_dim": n, "triplet_nnz": int(len(v)), "nnz_L": I.nnz_l, "flops_per_factor": I.flops_factor,
                       "flops_per_solve": I.flops_solve, "solves_per_step": NSOLVE, "parallelism": f"etree subtrees over {world} ranks, "
                       + ("top fronts replicated on the ranks beneath them (subtree-to-subcube), one RCCL all-reduce of the arena squares per bisection level"
                          if subcube else "one top replicated on every rank, one RCCL all-reduce of the top arena per factorisation")
                       + " inside libmi355x_kkt (no Python collective on the data path)", "subcube": subcube,
                       "supernodes": I.num_sn, "replicated_top_supernodes": int((own < 0).sum()), "num_neg": nneg, "scaled_residual": res},
            "same_workload_1gpu": {"ms_per_step": dt1 * 1e3, "value": flops_step / dt1 / 1e9, "speedup": dt1 / dt},
            "partition_model": partition_model(s, I, own, world),
            "range_local_collectives": cinfo["range_local"], "rccl_ranks_seen": cinfo["ranks_seen"], "subcomm": "on" if cinfo["range_local"] else "off",
            "communicator": cinfo,
            "roofline": roof,
        }
    if rank == 0:
        # the partition's bound with the exchange steps on top: T_N = T_1 / bound + the ring time of the arena squares (measured T_1 of this run)
        pm = line["partition_model"]; ex = pm["exchange"]["predicted_ms_per_factorisation"]
        t1 = line["same_workload_1gpu"]["ms_per_step"]
        pm["predicted_speedup_with_exchange"] = {kk: t1 / (t1 / pm["predicted_speedup_bound"] + ms) for kk, ms in ex.items()}
    # BASELINE configs[4]: an end-to-end Ipopt solve whose factorisations are shared by the N ranks -- one UNMODIFIED reference host process per
    # rank (oracle/_ref/ipopt_mi355x_driver, device route), communicator set up by the adapter itself (mi355x_comm rccl; shm on a shared device)
    if not getattr(args, "no_e2e", False):
        del ck, s
        e2e = e2e_multirank(rank, world, local, shared, dist)
        if rank == 0:
            line["e2e"] = e2e
    dist.barrier()
    if rank == 0:
        print(json.dumps(line))
    dist.destroy_process_group()


def e2e_multirank(rank, world, local, shared, dist, problem="MBndryCntrl1", size=700, golden="mbndry1_700"):
    """every rank runs the reference host on the configs[4] stand-in (MBndryCntrl1 N = 700: n = 492 800, m = 490 000, KKT dim 982 800) with the
    MI355X device route; rank 0 reports iterations / objective / Ipopt's own timers of every rank next to the reference CPU run's golden summary"""
    import subprocess
    drv = os.path.join(_ROOT, "oracle", "_ref", "ipopt_mi355x_driver")
    res = None
    if os.path.exists(drv):
        cmd = [drv, problem, str(size), "--solver", "mi355x-pd", "--quiet", "--set", "mi355x_nranks", str(world), "--set", "mi355x_rank", str(rank),
               "--set", "mi355x_device", str(local), "--set", "mi355x_comm", "shm" if shared else "rccl", "--set", "mi355x_subcube", "yes" if world > 2 else "no"]
        env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1", MI355X_KKT_SHM_TIMEOUT_S="120")
        env.setdefault("MI355X_KKT_JOB_ID", "bench-" + os.environ.get("MASTER_PORT", "0") + "-" + os.environ.get("TORCHELASTIC_RUN_ID", "0"))
        try:
            t0 = time.perf_counter()
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=180, cwd="/tmp", env=env).stdout      # (a run takes seconds; a rank whose peers never arrive is killed here -- the bench line survives with an error entry)
            j = json.loads(next(ln for ln in out.splitlines() if ln.startswith("DRIVER_SUMMARY"))[len("DRIVER_SUMMARY "):])
            res = {k: j[k] for k in ("iterations", "objective", "status", "PDSystemSolverTotal", "LinearSystemFactorization", "LinearSystemBackSolve",
                                     "LinearSystemSymbolicFactorization", "wall_total") if k in j}
            res["process_wall_s"] = time.perf_counter() - t0
        except Exception as e:      # a rank that fails must not take the bench line with it
            res = {"error": str(e)[:200]}
    allres = [None] * world
    dist.all_gather_object(allres, res)
    if rank != 0:
        return None
    if allres[0] is None:
        return {"skipped": "oracle/_ref/ipopt_mi355x_driver not built"}
    out = {"problem": f"ScalableProblems {problem} {size} (BASELINE configs[4] stand-in, KKT dim 982800)", "route": "mi355x-pd (device assembly + device-resident 8-block solver)",
           "communicator": "shm (ranks share one device)" if shared else "rccl", "ranks": allres}
    gpath = os.path.join(_ROOT, "tests", "golden", golden + ".summary")
    if os.path.exists(gpath):
        g = json.load(open(gpath))
        out["reference_cpu_run"] = {"iterations": g.get("iterations"), "objective": g.get("objective")}
        out["iterations_equal_on_every_rank"] = all(isinstance(r_, dict) and r_.get("iterations") == g.get("iterations") for r_ in allres)
    ok = [r_ for r_ in allres if isinstance(r_, dict) and "PDSystemSolverTotal" in r_]
    if ok:
        out["PDSystemSolverTotal_max_over_ranks"] = max(r_["PDSystemSolverTotal"] for r_ in ok)
        out["wall_total_max_over_ranks"] = max(r_["wall_total"] for r_ in ok)
    return out

"""-m 'not gpu': the C-ABI library loads on a machine without a GPU and exports every symbol that
include/mi355x_kkt.h declares; compute entry points fail LOUDLY (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

import ipopt_amd
from ipopt_amd import kkt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "mi355x_kkt.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355x_kkt_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(ipopt_amd.library_path())
    syms = header_symbols()
    assert len(syms) >= 19
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/mi355x_kkt.h but not exported"
    assert sorted(kkt.ABI_SYMBOLS) == syms


def test_ma97_compatible_exports():
    """route B2: the seven C symbols stock Ipopt dlsym()s from `hsllib`
    (reference IpMa97SolverInterface.cpp:308-314)."""
    lib = ctypes.CDLL(ipopt_amd.library_path())
    for s in ["ma97_default_control_d", "ma97_analyse_d", "ma97_factor_d", "ma97_factor_solve_d", "ma97_solve_d",
              "ma97_finalise_d", "ma97_free_akeep_d"]:
        assert hasattr(lib, s), s


def test_option_struct_layout_matches_header():
    o = kkt._Options()
    kkt.load_library().mi355x_kkt_default_options(ctypes.byref(o))
    assert (o.device, o.index_base, o.ordering, o.matching, o.scaling) == (-1, 1, 0, 1, 1)
    assert (o.nd_leaf, o.nemin, o.max_sn_cols) == (32, 8, 64)
    assert (o.pivtol, o.pivtolmax, o.small) == (1e-8, 1e-4, 1e-20)
    assert (o.use_graph, o.nranks, o.rank) == (1, 1, 0)
    assert (o.chain_group, o.solve_group, o.subcube) == (4, 0, 0)      # (subcube took the first of the reserved ints: same struct size)
    assert o.delay_rounds == 8                                          # (delay_rounds took the next reserved int)
    assert ctypes.sizeof(o) == ctypes.sizeof(kkt._Options) == 112 and o.smart_quality == 0
    assert ctypes.sizeof(kkt._Info) == 200                              # num_delayed / num_restructures live in what was reserved[0]


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a machine without a GPU")
def test_no_cpu_fallback_factor_fails_loudly():
    s = ipopt_amd.KKTSolver()
    r = np.array([1, 2, 2], dtype=np.int32); c = np.array([1, 1, 2], dtype=np.int32); v = np.array([2.0, 1.0, -1.0])
    s.initialize_structure(2, r, c, vals=v)        # symbolic analysis is host code and must work
    s.values()[:] = v
    with pytest.raises(ipopt_amd.KKTError, match="no HIP device|no CPU fallback|no usable HIP"):
        s.multi_solve(True, np.ones(2))


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a machine without a GPU")
def test_no_cpu_fallback_primal_dual_workspace_fails_loudly():
    """the 8-block entry points (mi355x_kkt_pd_*) need the device like factor/solve do: without one they return FATAL with a message,
    they do not compute anything on the host"""
    s = ipopt_amd.KKTSolver()
    n = 6
    s.initialize_structure(n, np.arange(1, n + 1), np.arange(1, n + 1), vals=np.ones(n))       # the analysis is host code: works
    with pytest.raises(ipopt_amd.KKTError, match="HIP device|no CPU fallback"):
        s.pd_define([n, 0, 0, 0, 0, 0, 0, 0], [], [], [], [], np.arange(1, n + 1), np.arange(1, n + 1), [0])
    with pytest.raises(ipopt_amd.KKTError):
        s.pd_solve_once(0, 1)
    with pytest.raises(ipopt_amd.KKTError):
        s.pd_residual(0, 1, 2, [0, 0, 0, 0])


def test_matching_scaling_is_a_maximum_product_scaling():
    """mi355x_kkt_matching_scaling (host, the job of MC64: Duff & Koster 2001): |s_i a_ij s_j| <= 1 everywhere, = 1 on a
    transversal (so every row and column of the scaled matrix has inf-norm exactly 1), also with zero diagonals and entries
    spread over 16 orders of magnitude -- the KKT shape MA97/SPRAL use it for."""
    import numpy as np
    from ipopt_amd import kkt
    from tests.support import kktgen
    lib = kkt.load_library()
    for gen in (lambda: kktgen.lukvl_like(400, seed=2, sigma_scale=1e3), lambda: kktgen.grid_kkt(9, 8, dof=2, ncon=2, seed=4, sigma_exp=8.0)):
        n, r, c, v, _ = gen()
        s = np.zeros(n); un = kkt.C.c_int(-1)
        assert lib.mi355x_kkt_matching_scaling(n, len(v), r.ctypes.data, c.ctypes.data, v.ctypes.data, 1, s.ctypes.data, kkt.C.byref(un)) == 0
        K = abs(kktgen.to_scipy(n, r, c, v)).multiply(s[:, None]).multiply(s[None, :]).tocsr()
        assert un.value == 0 and np.all(s > 0)
        assert K.max() <= 1.0 + 1e-10
        assert np.allclose(K.max(axis=1).toarray().ravel(), 1.0, rtol=1e-10)


def test_kernel_code_hash_and_the_cached_traffic_record():
    """bench.py emits a cached PMC traffic figure only for the sources or -- second key -- the kernel machine code it was measured with:
    the hash of a kernel's gfx950 code in the built library is well defined and stable, differs between kernels, is None for a name that
    matches nothing, and the committed record carries both keys."""
    import json
    import bench
    h1, h2 = bench.kernel_code_hash("k_big_schur"), bench.kernel_code_hash("k_big_schur")
    assert h1 is not None and re.fullmatch(r"[0-9a-f]{16}", h1) and h1 == h2
    assert bench.kernel_code_hash("k_fwd_chain") not in (None, h1)
    assert bench.kernel_code_hash("no_such_kernel") is None
    rec = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic_latest.json")))
    assert rec["workload"] == "synth_1e6" and rec["kernel"] == "k_big_schur" and rec["hbm_bytes_per_factorisation"] > 0
    assert re.fullmatch(r"[0-9a-f]{16}", rec["source_hash"]) and re.fullmatch(r"[0-9a-f]{16}", rec["kernel_code_hash"])


def test_committed_bench_lines_keep_the_contract():
    """the round's committed bench lines (profiles/rNN_bench_*.json, written by bench.py on the GPU box) carry every key of the driver's contract,
    the roofline object and the cpu_baseline object; the default workload is BASELINE configs[3]"""
    import glob
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "r0[3-9]_bench_*.json")))
    assert files
    for f in files:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
            assert k in j, (f, k)
        assert j["unit"] == "GFLOP/s" and j["dtype"] == "f64" and j["data"] == "synthetic" and j["higher_is_better"] is True and j["vs_baseline"] is None
        assert "workload" in j["config"] and "model" not in j["config"]
        r = j["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in r, (f, k)
        assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-9 * max(1.0, r["frac"])
        if "cpu_baseline" in j:
            for k in ("value", "unit", "cores", "kind", "sample"):
                assert k in j["cpu_baseline"], (f, k)
            assert j["cpu_baseline"]["kind"] in ("reference", "port")


def test_shared_memory_communicator_segment_is_created_and_discarded_without_a_device():
    """mi355x_kkt_comm_shm_id (rank 0 of `mi355x_comm shm`) is host code: the POSIX segment appears under /dev/shm with a name of its own per call,
    and mi355x_kkt_comm_shm_discard -- rank 0's way out when the id cannot be handed to the other ranks -- removes it again (nothing may be left behind:
    /dev/shm is memory).  The collectives themselves need a device: tests/test_multigpu_gpu.py, tests/test_e2e_multirank.py."""
    import ctypes as C
    import os
    lib = ipopt_amd.load_library()
    lib.mi355x_kkt_comm_shm_discard.restype = None
    ids = []
    for _ in range(2):
        buf = C.create_string_buffer(128)
        assert lib.mi355x_kkt_comm_shm_id(buf, 2) == 0
        name = buf.value.decode()
        assert name.startswith("/mi355x_kkt_") and os.path.exists("/dev/shm" + name)
        ids.append((buf, name))
    assert ids[0][1] != ids[1][1]
    for buf, name in ids:
        lib.mi355x_kkt_comm_shm_discard(buf)
        assert not os.path.exists("/dev/shm" + name)
        lib.mi355x_kkt_comm_shm_discard(buf)          # idempotent
    bad = C.create_string_buffer(128)
    assert lib.mi355x_kkt_comm_shm_id(bad, 0) != 0 and lib.mi355x_kkt_comm_shm_id(bad, 65) != 0

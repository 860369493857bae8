"""ctypes binding of include/mi355x_kkt.h.

Mirrors, method for method, the contract of the reference's SparseSymLinearSolverInterface
(reference src/Algorithm/LinearSolvers/IpSparseSymLinearSolverInterface.hpp:98-256):
``initialize_structure`` / ``values`` / ``multi_solve`` / ``number_of_neg_evals`` /
``increase_quality`` / ``provides_inertia`` -- same argument meaning, same status codes
(IpSymLinearSolver.hpp:19-33) -- so the parity tests read like the reference's adapters' callers
(IpTSymLinearSolver.cpp:159-312).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

STATUS = {0: "SUCCESS", 1: "SINGULAR", 2: "WRONG_INERTIA", 3: "CALL_AGAIN", 4: "FATAL_ERROR"}
SUCCESS, SINGULAR, WRONG_INERTIA, CALL_AGAIN, FATAL = 0, 1, 2, 3, 4
FMT_TRIPLET, FMT_CSR_UPPER = 0, 1


class KKTError(RuntimeError):
    pass


class _Options(C.Structure):
    _fields_ = [("device", C.c_int), ("index_base", C.c_int), ("ordering", C.c_int), ("matching", C.c_int),
                ("scaling", C.c_int), ("nd_leaf", C.c_int), ("nemin", C.c_int), ("max_sn_cols", C.c_int),
                ("pivtol", C.c_double), ("pivtolmax", C.c_double), ("small", C.c_double),
                ("refine_steps", C.c_int), ("use_graph", C.c_int), ("nranks", C.c_int), ("rank", C.c_int),
                ("verbose", C.c_int), ("leaf_cols", C.c_int), ("tree_merge", C.c_int), ("wide_panels", C.c_int), ("chain_group", C.c_int), ("solve_group", C.c_int), ("subcube", C.c_int), ("delay_rounds", C.c_int), ("smart_quality", C.c_int)]


class _Info(C.Structure):
    _fields_ = [("n", C.c_int), ("nnz_in", C.c_int), ("nnz_a", C.c_int), ("nnz_l", C.c_int64),
                ("flops_factor", C.c_int64), ("flops_solve", C.c_int64), ("bytes_factor", C.c_int64),
                ("bytes_solve", C.c_int64), ("sum_sn_rows", C.c_int64), ("cb_doubles", C.c_int64),
                ("num_sn", C.c_int), ("num_levels", C.c_int), ("maxfront", C.c_int), ("maxsupernode", C.c_int),
                ("num_pairs", C.c_int), ("num_neg", C.c_int), ("num_zero", C.c_int), ("num_two", C.c_int),
                ("num_small", C.c_int), ("num_big_fronts", C.c_int), ("time_analyse", C.c_double),
                ("time_factor_ms", C.c_double), ("time_solve_ms", C.c_double), ("pivtol", C.c_double), ("u_sensitive", C.c_int),
                ("num_fast_blocks", C.c_int), ("num_delayed", C.c_int), ("num_restructures", C.c_int), ("matching_ms", C.c_double), ("matching_rounds", C.c_int), ("matching_unmatched", C.c_int), ("reserved", C.c_double * 3)]


@dataclass
class KKTInfo:
    n: int; nnz_in: int; nnz_a: int; nnz_l: int; flops_factor: int; flops_solve: int; bytes_factor: int
    bytes_solve: int; sum_sn_rows: int; cb_doubles: int; num_sn: int; num_levels: int; maxfront: int
    maxsupernode: int; num_pairs: int; num_neg: int; num_zero: int; num_two: int; num_small: int
    num_big_fronts: int; time_analyse: float; time_factor_ms: float; time_solve_ms: float; pivtol: float; u_sensitive: int; num_fast_blocks: int; num_delayed: int; num_restructures: int
    matching_ms: float; matching_rounds: int; matching_unmatched: int


def library_path() -> str:
    # (MI355X_KKT_LIBRARY: development aid -- an A/B variant of the library built by `make variant`, see ipopt_amd/Makefile)
    return os.environ.get("MI355X_KKT_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmi355x_kkt.so")


_LIB = None
# int (*)(void* ctx, void* dptr, int64_t count, int dtype, void* hip_stream)  -- mi355x_kkt_allreduce_fn
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)
ALLREDUCE_RANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int)      # + rank_lo, nranks_in_range

# every symbol include/mi355x_kkt.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "mi355x_kkt_default_options", "mi355x_kkt_create", "mi355x_kkt_destroy", "mi355x_kkt_analyse",
    "mi355x_kkt_values_buffer", "mi355x_kkt_factor", "mi355x_kkt_refactor", "mi355x_kkt_solve",
    "mi355x_kkt_solve_device", "mi355x_kkt_solve_device2", "mi355x_kkt_set_pivtol", "mi355x_kkt_set_pivtolmax", "mi355x_kkt_increase_quality",
    "mi355x_kkt_get_info", "mi355x_kkt_last_error",
    "mi355x_kkt_get_symbolic", "mi355x_kkt_factor_local", "mi355x_kkt_top_arena", "mi355x_kkt_factor_top",
    "mi355x_kkt_solve_fwd_local", "mi355x_kkt_top_rhs", "mi355x_kkt_solve_top_and_bwd", "mi355x_kkt_profile",
    "mi355x_kkt_comm_unique_id", "mi355x_kkt_set_comm_rccl", "mi355x_kkt_comm_shm_id", "mi355x_kkt_comm_shm_discard", "mi355x_kkt_set_comm_shm", "mi355x_kkt_set_comm_callbacks", "mi355x_kkt_set_comm_range_callback", "mi355x_kkt_exchange_bytes", "mi355x_kkt_comm_plan", "mi355x_kkt_comm_info",
    "mi355x_kkt_set_scaling", "mi355x_kkt_get_scaling", "mi355x_kkt_ruiz_scaling", "mi355x_kkt_matching_scaling", "mi355x_kkt_zero_pivots", "mi355x_kkt_failed_pivots", "mi355x_kkt_delay_columns", "mi355x_kkt_set_delay_rounds", "mi355x_kkt_assembly_define", "mi355x_kkt_assembly_buffer", "mi355x_kkt_assembly_upload", "mi355x_kkt_factor_assembled",
    "mi355x_kkt_pd_define", "mi355x_kkt_pd_put_data", "mi355x_kkt_pd_put", "mi355x_kkt_pd_get", "mi355x_kkt_pd_solve_once", "mi355x_kkt_pd_residual",
]
KERNEL_KINDS = ["gather_scale", "front_wave", "front_lds64", "front_lds128", "big_assemble", "big_diag", "big_trsm", "big_schur",
                "stats", "solve_perm", "fwd_wave", "fwd_lds", "fwd_big", "bwd_wave", "bwd_lds", "bwd_big", "fwd_big_upd", "bwd_big_dot"]


def load_library():
    """dlopen the in-tree HIP library.  Fails loudly if it has not been built: there is no
    Python/numpy stand-in for the product path."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise KKTError(f"{path} is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(path)
    vp, ip, dp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)
    lib.mi355x_kkt_default_options.argtypes = [C.POINTER(_Options)]
    lib.mi355x_kkt_default_options.restype = None
    lib.mi355x_kkt_create.argtypes = [C.POINTER(vp), C.POINTER(_Options)]
    lib.mi355x_kkt_destroy.argtypes = [vp]
    lib.mi355x_kkt_destroy.restype = None
    lib.mi355x_kkt_analyse.argtypes = [vp, C.c_int, C.c_int, vp, vp, C.c_int, vp]
    lib.mi355x_kkt_values_buffer.argtypes = [vp]
    lib.mi355x_kkt_values_buffer.restype = dp
    lib.mi355x_kkt_factor.argtypes = [vp, vp, ip, ip]
    lib.mi355x_kkt_refactor.argtypes = [vp, ip, ip]
    lib.mi355x_kkt_solve.argtypes = [vp, C.c_int, vp, C.c_int]
    lib.mi355x_kkt_solve_device.argtypes = [vp, C.c_int, vp, C.c_int]
    lib.mi355x_kkt_solve_device2.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_int]
    lib.mi355x_kkt_set_pivtol.argtypes = [vp, C.c_double]
    lib.mi355x_kkt_set_pivtolmax.argtypes = [vp, C.c_double]
    lib.mi355x_kkt_increase_quality.argtypes = [vp, dp]
    lib.mi355x_kkt_get_info.argtypes = [vp, C.POINTER(_Info)]
    lib.mi355x_kkt_last_error.argtypes = [vp]
    lib.mi355x_kkt_last_error.restype = C.c_char_p
    lib.mi355x_kkt_get_symbolic.argtypes = [vp, C.c_int, vp, C.c_int64]
    lib.mi355x_kkt_profile.argtypes = [vp, C.c_int, vp, vp, C.c_int]
    lib.mi355x_kkt_factor_local.argtypes = [vp, vp]
    lib.mi355x_kkt_top_arena.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int64)]
    lib.mi355x_kkt_factor_top.argtypes = [vp, ip, ip]
    lib.mi355x_kkt_solve_fwd_local.argtypes = [vp, vp]
    lib.mi355x_kkt_top_rhs.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int64)]
    lib.mi355x_kkt_solve_top_and_bwd.argtypes = [vp, vp]
    lib.mi355x_kkt_comm_unique_id.argtypes = [vp]
    lib.mi355x_kkt_set_scaling.argtypes = [vp, C.c_int, vp]
    lib.mi355x_kkt_get_scaling.argtypes = [vp, vp]
    lib.mi355x_kkt_ruiz_scaling.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, C.c_int, vp]
    lib.mi355x_kkt_matching_scaling.argtypes = [C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, ip]
    lib.mi355x_kkt_zero_pivots.argtypes = [vp, vp, C.c_int, ip]
    lib.mi355x_kkt_failed_pivots.argtypes = [vp, vp, C.c_int, ip]
    lib.mi355x_kkt_delay_columns.argtypes = [vp, vp, C.c_int, ip]
    lib.mi355x_kkt_set_delay_rounds.argtypes = [vp, C.c_int]
    lib.mi355x_kkt_assembly_define.argtypes = [vp, C.c_int, vp, vp]
    lib.mi355x_kkt_assembly_buffer.argtypes = [vp, C.c_int]
    lib.mi355x_kkt_assembly_buffer.restype = dp
    lib.mi355x_kkt_assembly_upload.argtypes = [vp, C.c_int]
    lib.mi355x_kkt_factor_assembled.argtypes = [vp, vp, vp, ip, ip]
    lib.mi355x_kkt_pd_define.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int]
    lib.mi355x_kkt_pd_put_data.argtypes = [vp, vp]
    lib.mi355x_kkt_pd_put.argtypes = [vp, C.c_int, vp]
    lib.mi355x_kkt_pd_get.argtypes = [vp, C.c_int, vp]
    lib.mi355x_kkt_pd_solve_once.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_double]
    lib.mi355x_kkt_pd_residual.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.mi355x_kkt_set_comm_rccl.argtypes = [vp, vp]
    lib.mi355x_kkt_comm_shm_id.argtypes = [vp, C.c_int]
    lib.mi355x_kkt_set_comm_shm.argtypes = [vp, vp]
    lib.mi355x_kkt_set_comm_callbacks.argtypes = [vp, ALLREDUCE_FN, vp]
    lib.mi355x_kkt_set_comm_range_callback.argtypes = [vp, ALLREDUCE_RANGE_FN]
    lib.mi355x_kkt_exchange_bytes.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.mi355x_kkt_comm_plan.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, ip]
    lib.mi355x_kkt_comm_info.argtypes = [vp, ip, ip, ip, ip]
    _LIB = lib
    return lib


class KKTSolver:
    """One solver instance == one ``mi355x_kkt_handle``."""

    def __init__(self, **opts):
        self.lib = load_library()
        o = _Options()
        self.lib.mi355x_kkt_default_options(C.byref(o))
        for k, v in opts.items():
            if not hasattr(o, k):
                raise KKTError(f"unknown option {k}")
            setattr(o, k, v)
        self._opts = o
        self._h = C.c_void_p()
        if self.lib.mi355x_kkt_create(C.byref(self._h), C.byref(o)) != 0:
            raise KKTError("mi355x_kkt_create failed")
        self._nnz = 0
        self._n = 0
        self._neg = 0
        self.pivtol = o.pivtol
        self.pivtolmax = o.pivtolmax

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.mi355x_kkt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self) -> str:
        return self.lib.mi355x_kkt_last_error(self._h).decode()

    # --- InitializeStructure (IpSparseSymLinearSolverInterface.hpp:139) ---
    def initialize_structure(self, n, row, col, fmt=FMT_TRIPLET, vals=None) -> int:
        row = np.ascontiguousarray(row, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        nnz = int(col.shape[0])
        v = None
        if vals is not None:
            v = np.ascontiguousarray(vals, dtype=np.float64)
            assert v.shape[0] == nnz
        st = self.lib.mi355x_kkt_analyse(self._h, int(n), nnz, row.ctypes.data, col.ctypes.data, int(fmt),
                                         v.ctypes.data if v is not None else None)
        if st != 0:
            raise KKTError("analyse: " + self.last_error())
        self._nnz, self._n = nnz, int(n)
        return st

    # --- GetValuesArrayPtr (hpp:155) ---
    def values(self) -> np.ndarray:
        p = self.lib.mi355x_kkt_values_buffer(self._h)
        if not p:
            raise KKTError("values_buffer: " + self.last_error())
        return np.ctypeslib.as_array(p, shape=(max(self._nnz, 1),))[: self._nnz]

    # --- MultiSolve (hpp:190) ---
    def multi_solve(self, new_matrix: bool, rhs: np.ndarray | None, check_neg_evals=False, number_of_neg_evals=0) -> int:
        if new_matrix or self._pending_refactor():
            neg, zero = C.c_int(0), C.c_int(0)
            if new_matrix:
                st = self.lib.mi355x_kkt_factor(self._h, None, C.byref(neg), C.byref(zero))
            else:
                st = self.lib.mi355x_kkt_refactor(self._h, C.byref(neg), C.byref(zero))
            self._refactor = False
            if st == FATAL:
                raise KKTError("factor: " + self.last_error())
            self._neg = neg.value
            if st == SINGULAR:
                return SINGULAR
            if check_neg_evals and self._neg != number_of_neg_evals:
                return WRONG_INERTIA
        if rhs is not None and rhs.size:
            assert rhs.dtype == np.float64 and rhs.flags.c_contiguous
            nrhs = 1 if rhs.ndim == 1 else rhs.shape[0]
            st = self.lib.mi355x_kkt_solve(self._h, nrhs, rhs.ctypes.data, self._n)
            if st != 0:
                raise KKTError("solve: " + self.last_error())
        return SUCCESS

    def factor_device(self, dvals_ptr: int):
        neg, zero = C.c_int(0), C.c_int(0)
        st = self.lib.mi355x_kkt_factor(self._h, C.c_void_p(dvals_ptr), C.byref(neg), C.byref(zero))
        if st == FATAL:
            raise KKTError("factor: " + self.last_error())
        self._neg = neg.value
        return st, neg.value, zero.value

    def solve_device(self, drhs_ptr: int, nrhs=1):
        st = self.lib.mi355x_kkt_solve_device(self._h, nrhs, C.c_void_p(drhs_ptr), self._n)
        if st != 0:
            raise KKTError("solve_device: " + self.last_error())

    def solve_device2(self, db_ptr: int, dx_ptr: int, nrhs=1):
        st = self.lib.mi355x_kkt_solve_device2(self._h, nrhs, C.c_void_p(db_ptr), self._n, C.c_void_p(dx_ptr), self._n)
        if st != 0:
            raise KKTError("solve_device2: " + self.last_error())

    # --- device-side value assembly: segments  values[off + i] = scale * src[i] + shift ---
    def assembly_define(self, lengths):
        ln = np.asarray(lengths, dtype=np.int64)
        off = np.concatenate([[0], np.cumsum(ln)[:-1]]).astype(np.int64)
        if self.lib.mi355x_kkt_assembly_define(self._h, len(ln), off.ctypes.data, ln.ctypes.data) != 0:
            raise KKTError("assembly_define: " + self.last_error())
        self._seg_len = ln

    def assembly_set(self, seg, values):
        p = self.lib.mi355x_kkt_assembly_buffer(self._h, int(seg))
        n = int(self._seg_len[seg])
        if n:
            np.ctypeslib.as_array(p, shape=(n,))[:] = values
        if self.lib.mi355x_kkt_assembly_upload(self._h, int(seg)) != 0:
            raise KKTError("assembly_upload: " + self.last_error())

    def factor_assembled(self, scale, shift):
        sc = np.ascontiguousarray(scale, dtype=np.float64); sh = np.ascontiguousarray(shift, dtype=np.float64)
        neg, zero = C.c_int(0), C.c_int(0)
        st = self.lib.mi355x_kkt_factor_assembled(self._h, sc.ctypes.data, sh.ctypes.data, C.byref(neg), C.byref(zero))
        if st == FATAL:
            raise KKTError("factor_assembled: " + self.last_error())
        self._neg = neg.value
        return st, neg.value, zero.value

    # --- the 8-block primal-dual system on the device: vectors are lists of 8 arrays x|s|y_c|y_d|z_L|z_U|v_L|v_U ---
    def pd_define(self, dims8, idx_xl, idx_xu, idx_sl, idx_su, irn, jcn, segs):
        d = np.ascontiguousarray(dims8, dtype=np.int32)
        ix = [np.ascontiguousarray(a, dtype=np.int32) for a in (idx_xl, idx_xu, idx_sl, idx_su)]
        r = np.ascontiguousarray(irn, dtype=np.int32); c = np.ascontiguousarray(jcn, dtype=np.int32); sg = np.ascontiguousarray(segs, dtype=np.int32)
        if self.lib.mi355x_kkt_pd_define(self._h, d.ctypes.data, ix[0].ctypes.data, ix[1].ctypes.data, ix[2].ctypes.data, ix[3].ctypes.data,
                                         r.ctypes.data, c.ctypes.data, sg.ctypes.data, len(sg)) != 0:
            raise KKTError("pd_define: " + self.last_error())
        self._pd_len = [int(v) for v in d]

    @staticmethod
    def _ptr_array(arrs):
        keep = [np.ascontiguousarray(a, dtype=np.float64) for a in arrs]
        return (C.c_void_p * 8)(*[a.ctypes.data if a.size else None for a in keep]), keep

    def pd_put_data(self, data8):
        pa, keep = self._ptr_array(data8)
        if self.lib.mi355x_kkt_pd_put_data(self._h, pa) != 0:
            raise KKTError("pd_put_data: " + self.last_error())

    def pd_put(self, vec, blocks8):
        pa, keep = self._ptr_array(blocks8)
        if self.lib.mi355x_kkt_pd_put(self._h, int(vec), pa) != 0:
            raise KKTError("pd_put: " + self.last_error())

    def pd_get(self, vec):
        out = [np.zeros(n) for n in self._pd_len]
        pa = (C.c_void_p * 8)(*[a.ctypes.data if a.size else None for a in out])
        if self.lib.mi355x_kkt_pd_get(self._h, int(vec), pa) != 0:
            raise KKTError("pd_get: " + self.last_error())
        return out

    def pd_solve_once(self, rhs, res, alpha=1.0, beta=0.0):
        if self.lib.mi355x_kkt_pd_solve_once(self._h, int(rhs), int(res), float(alpha), float(beta)) != 0:
            raise KKTError("pd_solve_once: " + self.last_error())

    def pd_residual(self, rhs, res, resid, deltas4):
        d = np.ascontiguousarray(deltas4, dtype=np.float64); nr = np.zeros(3)
        if self.lib.mi355x_kkt_pd_residual(self._h, int(rhs), int(res), int(resid), d.ctypes.data, nr.ctypes.data) != 0:
            raise KKTError("pd_residual: " + self.last_error())
        return nr

    # --- multi-GPU communicator (include/mi355x_kkt.h): after one of these, multi_solve / factor_device / solve_device* of a
    #     handle created with nranks > 1 run the distributed sequence inside the library ---
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        if load_library().mi355x_kkt_comm_unique_id(buf) != 0:
            raise KKTError("comm_unique_id: librccl.so not loadable / ncclGetUniqueId failed")
        return buf.raw

    def set_comm_rccl(self, unique_id: bytes):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        if self.lib.mi355x_kkt_set_comm_rccl(self._h, buf) != 0:
            raise KKTError("set_comm_rccl: " + self.last_error())

    @staticmethod
    def comm_shm_id(nranks: int) -> bytes:
        """rank 0: create the shared-memory segment of a host-staged communicator for ranks of one node (they may share a device)"""
        buf = C.create_string_buffer(128)
        if load_library().mi355x_kkt_comm_shm_id(buf, int(nranks)) != 0:
            raise KKTError("comm_shm_id: shm_open / mmap failed")
        return buf.raw

    def set_comm_shm(self, shm_id: bytes):
        buf = C.create_string_buffer(bytes(shm_id), 128)
        if self.lib.mi355x_kkt_set_comm_shm(self._h, buf) != 0:
            raise KKTError("set_comm_shm: " + self.last_error())

    def set_comm_callback(self, fn, fn_range=None):
        """fn(dptr: int, count: int, dtype: int (0 fp64, 1 int32), hip_stream: int) -> None; must leave the buffer summed over the ranks.
        fn_range(dptr, count, dtype, hip_stream, rank_lo, nranks_in_range) (optional): the same over a range of ranks only"""
        def _wrap(f):
            def _cb(ctx, *a):
                try:
                    f(*[int(x or 0) for x in a])
                    return 0
                except Exception:          # never let an exception cross the C ABI
                    import traceback; traceback.print_exc()
                    return 1
            return _cb
        self._comm_cb = ALLREDUCE_FN(_wrap(fn))       # keep the thunks alive as long as the handle
        if self.lib.mi355x_kkt_set_comm_callbacks(self._h, self._comm_cb, None) != 0:
            raise KKTError("set_comm_callbacks: " + self.last_error())
        if fn_range is not None:
            self._comm_range_cb = ALLREDUCE_RANGE_FN(_wrap(fn_range))
            if self.lib.mi355x_kkt_set_comm_range_callback(self._h, self._comm_range_cb) != 0:
                raise KKTError("set_comm_range_callback: " + self.last_error())

    def exchange_bytes(self):
        """(bytes of all arena squares, bytes of all top right-hand sides) of the current multi-GPU structure"""
        a, t = C.c_int64(0), C.c_int64(0)
        if self.lib.mi355x_kkt_exchange_bytes(self._h, C.byref(a), C.byref(t)) != 0:
            raise KKTError("exchange_bytes: no device-side set-up")
        return a.value, t.value

    def comm_plan(self, rank, range_local=True):
        """host only: the collectives rank `rank` issues for one factorisation + one solve (and the ncclCommSplit calls before them) as an (k, 6) int array
        {what, step, colour, range size, count, dtype} -- include/mi355x_kkt.h mi355x_kkt_comm_plan.  Needs the analysis, not a device."""
        cnt = C.c_int(0)
        if self.lib.mi355x_kkt_comm_plan(self._h, int(rank), int(bool(range_local)), None, 0, C.byref(cnt)) != 0:
            raise KKTError("comm_plan: " + self.last_error())
        out = np.zeros((max(cnt.value, 1), 6), dtype=np.int32)
        if self.lib.mi355x_kkt_comm_plan(self._h, int(rank), int(bool(range_local)), out.ctypes.data, cnt.value, C.byref(cnt)) != 0:
            raise KKTError("comm_plan: " + self.last_error())
        return out[:cnt.value]

    def comm_info(self):
        """{kind: 0 none / 1 callbacks / 2 rccl, ranks_seen (ncclCommCount), range_local, exchange_steps} of the communicator in use"""
        k, r, l, e = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        if self.lib.mi355x_kkt_comm_info(self._h, C.byref(k), C.byref(r), C.byref(l), C.byref(e)) != 0:
            raise KKTError("comm_info: " + self.last_error())
        return {"kind": ("none", "callbacks", "rccl")[k.value], "ranks_seen": r.value, "range_local": bool(l.value), "exchange_steps": e.value}

    _refactor = False

    def _pending_refactor(self):
        return self._refactor

    def number_of_neg_evals(self) -> int:
        return self._neg

    # --- IncreaseQuality (hpp:220): u <- min(umax, u^0.75), as MA97/SPRAL/MA27 adapters do; False when u is at its
    #     maximum OR the last factorisation found no pivot decision that depends on u (nothing a refactorisation could change) ---
    def increase_quality(self) -> bool:
        u = C.c_double(0.0)
        if self.lib.mi355x_kkt_increase_quality(self._h, C.byref(u)) == 0:
            return False
        self.pivtol = u.value
        self._refactor = True
        return True

    def set_pivtol(self, u: float):
        if self.lib.mi355x_kkt_set_pivtol(self._h, float(u)) != 0:
            raise KKTError("set_pivtol: " + self.last_error())
        self.pivtol = float(u)

    @staticmethod
    def provides_inertia() -> bool:
        return True

    def set_scaling(self, mode: int, factors=None):
        f = None if factors is None else np.ascontiguousarray(factors, dtype=np.float64)
        if self.lib.mi355x_kkt_set_scaling(self._h, int(mode), f.ctypes.data if f is not None else None) != 0:
            raise KKTError("set_scaling: " + self.last_error())

    def get_scaling(self) -> np.ndarray:
        out = np.zeros(max(self._n, 1))
        if self.lib.mi355x_kkt_get_scaling(self._h, out.ctypes.data) != 0:
            raise KKTError("get_scaling: " + self.last_error())
        return out[: self._n]

    # --- DetermineDependentRows support (hpp:240-255): columns with a zero pivot in the last factorisation (caller's index base) ---
    def zero_pivots(self) -> np.ndarray:
        cnt = C.c_int(0)
        if self.lib.mi355x_kkt_zero_pivots(self._h, None, 0, C.byref(cnt)) != 0:
            raise KKTError("zero_pivots: " + self.last_error())
        out = np.zeros(max(cnt.value, 1), dtype=np.int32)
        self.lib.mi355x_kkt_zero_pivots(self._h, out.ctypes.data, cnt.value, C.byref(cnt))
        return out[:cnt.value]

    # --- delayed pivoting across fronts (include/mi355x_kkt.h): what the last factorisation had to force, and the structural edit ---
    def failed_pivots(self) -> np.ndarray:
        cnt = C.c_int(0)
        if self.lib.mi355x_kkt_failed_pivots(self._h, None, 0, C.byref(cnt)) != 0:
            raise KKTError("failed_pivots: " + self.last_error())
        out = np.zeros(max(cnt.value, 1), dtype=np.int32)
        self.lib.mi355x_kkt_failed_pivots(self._h, out.ctypes.data, cnt.value, C.byref(cnt))
        return out[:cnt.value]

    def delay_columns(self, cols) -> int:
        """move the columns (caller's index base) to their parent fronts; returns how many moved"""
        c = np.ascontiguousarray(cols, dtype=np.int32)
        moved = C.c_int(0)
        if self.lib.mi355x_kkt_delay_columns(self._h, c.ctypes.data, int(c.shape[0]), C.byref(moved)) != 0:
            raise KKTError("delay_columns: " + self.last_error())
        return moved.value

    def set_delay_rounds(self, rounds: int):
        if self.lib.mi355x_kkt_set_delay_rounds(self._h, int(rounds)) != 0:
            raise KKTError("set_delay_rounds: rounds >= 0")

    def info(self) -> KKTInfo:
        i = _Info()
        self.lib.mi355x_kkt_get_info(self._h, C.byref(i))
        return KKTInfo(**{f: getattr(i, f) for f in KKTInfo.__dataclass_fields__})

    def profile(self, reps=1) -> dict:
        """{kernel kind: (total device ms over reps, launches over reps)} from hip events around every launch."""
        ms = np.zeros(len(KERNEL_KINDS)); ln = np.zeros(len(KERNEL_KINDS), dtype=np.int32)
        if self.lib.mi355x_kkt_profile(self._h, int(reps), ms.ctypes.data, ln.ctypes.data, len(KERNEL_KINDS)) != 0:
            raise KKTError("profile: " + self.last_error())
        return {k: (float(ms[i]), int(ln[i])) for i, k in enumerate(KERNEL_KINDS)}

    def symbolic(self, what: int, size: int) -> np.ndarray:
        out = np.empty(max(size, 1), dtype=np.int32)
        if self.lib.mi355x_kkt_get_symbolic(self._h, what, out.ctypes.data, out.shape[0]) != 0:
            raise KKTError("get_symbolic: " + self.last_error())
        return out[:size]

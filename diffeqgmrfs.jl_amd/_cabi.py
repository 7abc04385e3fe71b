"""ctypes binding of libgmrf_hip.so (include/gmrf_hip.h).

This is the Python twin of the Julia `ccall` shim in julia/DiffEqGMRFsHIP.jl: plain
pointers and sizes only.  The library is the product; there is NO CPU fallback -- if the
shared object is missing or no GPU is visible the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GMRF_HIP_LIBRARY") or os.path.join(_HERE, "csrc", "libgmrf_hip.so")   # (override: A/B of two builds, tools/)

GMRF_OK = 0
ERR_NOT_SPD, ERR_BAD_SHAPE, ERR_BAND, ERR_HIP, ERR_NO_FACTOR, ERR_NO_DEVICE, ERR_ALLOC, ERR_RCCL = -1, -2, -3, -4, -5, -6, -7, -8
SOLVE_FULL, SOLVE_FORWARD, SOLVE_BACKWARD = 0, 1, 2
VAR_EXACT, VAR_RBMC, VAR_MC = 0, 1, 2
BLOCK_L, BLOCK_C, BLOCK_LINV = 0, 1, 2

# every symbol include/gmrf_hip.h declares (tests check that the library exports them all)
EXPORTS = [
    "gmrf_bt_create", "gmrf_bt_destroy", "gmrf_last_error", "gmrf_version",
    "gmrf_bt_factor_csc", "gmrf_bt_factor_blocks", "gmrf_bt_refactor_values",
    "gmrf_bt_solve", "gmrf_bt_sample", "gmrf_bt_posterior", "gmrf_bt_normals", "gmrf_bt_marginal_var",
    "gmrf_bt_var_accumulate", "gmrf_bt_logdet", "gmrf_bt_get_block", "gmrf_bt_factor_buffer",
    "gmrf_bt_adopt_shape", "gmrf_bt_adopt_commit", "gmrf_bt_adopt_layout", "gmrf_bt_get_layout", "gmrf_bt_block_range",
    "gmrf_bt_set_keep_l", "gmrf_bt_storage_bytes", "gmrf_bt_set_storage", "gmrf_bt_factor_begin_csc",
    "gmrf_bt_factor_step_async", "gmrf_bt_factor_end", "gmrf_bt_stats",
    "gmrf_bt_set_profiling", "gmrf_bt_set_eager", "gmrf_bt_synchronize", "gmrf_bt_set_batch", "gmrf_bt_select_problem",
    "gmrf_bt_marginal_var_batch", "gmrf_bt_export_size", "gmrf_bt_export_factor", "gmrf_bt_import_factor",
    "gmrf_comm_unique_id", "gmrf_comm_create", "gmrf_comm_destroy", "gmrf_comm_bcast_host", "gmrf_comm_allreduce_sum",
    "gmrf_bt_bcast_blocks_async", "gmrf_bt_allgather_blocks_async", "gmrf_comm_wait", "gmrf_comm_bytes", "gmrf_streams_create", "gmrf_streams_destroy",
    "gmrf_bt_packed_size", "gmrf_bt_pack_blocks_async", "gmrf_bt_unpack_blocks_async",
    "gmrf_csr_create", "gmrf_csr_destroy", "gmrf_spmm", "gmrf_spmm_rows", "gmrf_spmm_async", "gmrf_spmm_rows_async",
    "gmrf_darcy_p1_create", "gmrf_darcy_p2_create", "gmrf_darcy_p1_destroy", "gmrf_darcy_p1_pattern", "gmrf_darcy_p1_assemble",
    "gmrf_burgers_p1_create", "gmrf_burgers_p2_create", "gmrf_burgers_p1_destroy", "gmrf_burgers_p1_pattern", "gmrf_burgers_p1_tangent",
    "gmrf_shallow_water_p1_create", "gmrf_shallow_water_p1_destroy", "gmrf_shallow_water_p1_pattern", "gmrf_shallow_water_p1_qpoints",
    "gmrf_shallow_water_p1_assemble", "gmrf_shallow_water_p1_operators",
    "gmrf_assemble_create", "gmrf_assemble_destroy", "gmrf_assemble_pattern", "gmrf_assemble_precision", "gmrf_assemble_rhs",
    "gmrf_test_gemm", "gmrf_test_gemm_rate", "gmrf_test_gemm_shapes", "gmrf_test_potrf_tile", "gmrf_test_potrf_block", "gmrf_test_tile_timing", "gmrf_test_persist_stamps", "gmrf_test_persist_aborts", "gmrf_test_persist_budget", "gmrf_test_clock_probe_start", "gmrf_test_clock_probe_finish",
    "gmrf_test_mfma_f64_rate", "gmrf_test_hbm_rate", "gmrf_test_microbench", "gmrf_test_symbolic_csc",
]


class SparseBlock(C.Structure):
    _fields_ = [("nnz", C.c_int64), ("ptr", C.c_void_p), ("idx", C.c_void_p), ("val", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [
        ("factor_ms", C.c_double), ("solve_ms", C.c_double), ("sample_ms", C.c_double),
        ("factor_flops", C.c_double), ("sweep_bytes", C.c_double), ("sweep_ms", C.c_double),
        ("n", C.c_int64), ("n_blocks", C.c_int64), ("block_size", C.c_int64),
        ("block_size_padded", C.c_int64), ("factor_bytes", C.c_int64),
        ("kernel_ms", C.c_double * 24), ("kernel_work", C.c_double * 24), ("kernel_launches", C.c_int64 * 24),
        ("sweep_bytes_streamed", C.c_double),
        ("persist_route", C.c_int32), ("persist_aborts", C.c_int32), ("persist_cus", C.c_int32), ("persist_refused", C.c_int32),
        ("sweep_persist", C.c_int32), ("sweep_persist_launches", C.c_int32),
    ]


class GmrfError(RuntimeError):
    def __init__(self, status: int, msg: str, info: int = 0):
        super().__init__(f"libgmrf_hip status {status}: {msg}")
        self.status = status
        self.info = info


class NotPositiveDefinite(GmrfError):
    """Julia's PosDefException(info): `.info` is the failing block (1-based)."""


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen the in-tree library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C diffeqgmrfs.jl_amd/csrc` (the HIP library is the only compute path)")
    # PyTorch wheels bundle their own libamdhip64 (same SONAME as /opt/rocm's).  Two HIP
    # runtimes in one process cannot both own the GPU, so when torch is installed it is
    # imported FIRST: the loader then binds this library to the runtime torch already
    # mapped, and device pointers / streams can be shared between the two.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the C ABI itself
        pass
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double
    P = C.POINTER
    sigs = {
        "gmrf_bt_create": [i32, vp, P(vp)],
        "gmrf_bt_destroy": [vp],
        "gmrf_version": [],
        "gmrf_bt_factor_csc": [vp, i64, i64, vp, vp, vp, i32, P(i32)],
        "gmrf_bt_factor_blocks": [vp, i64, i64, vp, vp, i32, i32, P(i32)],
        "gmrf_bt_refactor_values": [vp, vp, P(i32)],
        "gmrf_bt_solve": [vp, vp, vp, i64, i64, i64, i32],
        "gmrf_bt_sample": [vp, u64, i64, i64, vp, vp, vp, i64],
        "gmrf_bt_posterior": [vp, vp, u64, i64, i64, vp, vp, i64],
        "gmrf_bt_normals": [vp, u64, i64, i64, vp, i64],
        "gmrf_bt_marginal_var": [vp, i32, i64, u64, vp, vp],
        "gmrf_bt_var_accumulate": [vp, i32, i64, i64, u64, vp, vp],
        "gmrf_bt_logdet": [vp, P(dbl)],
        "gmrf_bt_get_block": [vp, i32, i64, vp, i64],
        "gmrf_bt_factor_buffer": [vp, i32, P(vp), P(i64)],
        "gmrf_bt_adopt_shape": [vp, i64, i64],
        "gmrf_bt_storage_bytes": [i64, i64, i64, P(i64), P(i64), P(i64)],
        "gmrf_bt_set_storage": [vp, i64, i64, i64, vp, vp, vp],
        "gmrf_bt_adopt_commit": [vp, i32],
        "gmrf_bt_adopt_layout": [vp, i64, i64, vp, i64],
        "gmrf_bt_get_layout": [vp, vp, i64, P(i64)],
        "gmrf_bt_block_range": [vp, i32, i64, i64, P(i64), P(i64), P(i64)],
        "gmrf_bt_set_keep_l": [vp, i32],
        "gmrf_comm_unique_id": [vp],
        "gmrf_comm_create": [i32, i32, i32, vp, P(vp)],
        "gmrf_comm_destroy": [vp],
        "gmrf_comm_bcast_host": [vp, vp, i64, i32],
        "gmrf_comm_allreduce_sum": [vp, vp, vp, i64],
        "gmrf_bt_bcast_blocks_async": [vp, vp, i32, i64, i64, i32],
        "gmrf_bt_allgather_blocks_async": [vp, vp, vp, i64, i64],
        "gmrf_comm_wait": [vp, vp],
        "gmrf_comm_bytes": [vp, i32, P(dbl)],
        "gmrf_bt_packed_size": [vp, i64, i64, P(i64)],
        "gmrf_bt_pack_blocks_async": [vp, i64, i64, vp],
        "gmrf_bt_unpack_blocks_async": [vp, i64, i64, vp],
        "gmrf_streams_create": [i32, i32, vp, vp],
        "gmrf_streams_destroy": [i32, i32, vp],
        "gmrf_bt_factor_begin_csc": [vp, i64, i64, vp, vp, vp, i32],
        "gmrf_bt_factor_step_async": [vp, i64, i64],
        "gmrf_bt_factor_end": [vp, P(i32)],
        "gmrf_bt_stats": [vp, P(Stats)],
        "gmrf_bt_set_profiling": [vp, i32],
        "gmrf_bt_set_eager": [vp, i32],
        "gmrf_bt_synchronize": [vp],
        "gmrf_bt_set_batch": [vp, i64],
        "gmrf_bt_select_problem": [vp, i64],
        "gmrf_csr_create": [i32, vp, i64, i64, vp, vp, vp, i32, i32, P(vp)],
        "gmrf_bt_marginal_var_batch": [vp, i32, i64, u64, vp, vp, vp],
        "gmrf_bt_export_size": [vp, P(i64)],
        "gmrf_bt_export_factor": [vp, vp, i64],
        "gmrf_bt_import_factor": [vp, vp, i64],
        "gmrf_csr_destroy": [vp],
        "gmrf_assemble_create": [i32, vp, i64, vp, vp, i64, vp, vp, i32, P(vp)],
        "gmrf_assemble_destroy": [vp],
        "gmrf_assemble_pattern": [vp, P(i64), P(i64), vp, vp, i32],
        "gmrf_assemble_precision": [vp, vp, vp, dbl, vp],
        "gmrf_assemble_rhs": [vp, vp, vp, vp, vp, dbl, vp],
        "gmrf_darcy_p1_create": [i32, vp, i64, i64, P(vp)],
        "gmrf_darcy_p2_create": [i32, vp, i64, i64, P(vp)],
        "gmrf_darcy_p1_destroy": [vp],
        "gmrf_darcy_p1_pattern": [vp, P(i64), vp, vp, i32],
        "gmrf_darcy_p1_assemble": [vp, vp, i64, dbl, vp, vp],
        "gmrf_burgers_p1_create": [i32, vp, i64, i64, dbl, dbl, P(vp)],
        "gmrf_burgers_p2_create": [i32, vp, i64, i64, dbl, dbl, P(vp)],
        "gmrf_burgers_p1_destroy": [vp],
        "gmrf_burgers_p1_pattern": [vp, P(i64), vp, vp, i32],
        "gmrf_burgers_p1_tangent": [vp, vp, vp, vp],
        "gmrf_shallow_water_p1_create": [i32, vp, i64, i64, P(vp)],
        "gmrf_shallow_water_p1_destroy": [vp],
        "gmrf_shallow_water_p1_pattern": [vp, i32, P(i64), vp, vp, i32],
        "gmrf_shallow_water_p1_qpoints": [vp, vp],
        "gmrf_shallow_water_p1_assemble": [vp, vp, dbl, dbl, dbl, vp, vp, vp, vp],
        "gmrf_shallow_water_p1_operators": [vp, vp, vp, vp, vp, dbl, dbl, dbl, vp, vp, vp, vp],
        "gmrf_spmm": [vp, vp, vp, i64, i64, i64],
        "gmrf_spmm_async": [vp, vp, vp, i64, i64, i64],
        "gmrf_spmm_rows_async": [vp, vp, vp, i64, i64, i64],
        "gmrf_spmm_rows": [vp, vp, vp, i64, i64, i64],
        "gmrf_test_gemm": [i32, i64, i64, i64, i32, i32, i32, i32, dbl, vp, i64, vp, i64, dbl, vp, i64],
        "gmrf_test_gemm_shapes": [vp, P(dbl), i64, P(i64)],
        "gmrf_test_gemm_rate": [i32, i64, i64, i64, i32, i32, i32, i32, i32, i32, P(dbl)],
        "gmrf_test_potrf_tile": [i32, vp, vp, P(i32)],
        "gmrf_test_potrf_block": [i32, i64, vp, vp, P(i32)],
        "gmrf_test_tile_timing": [vp, i32],
        "gmrf_test_persist_stamps": [vp, i32],
        "gmrf_test_persist_aborts": [vp, vp],
        "gmrf_test_persist_budget": [i32, i32, vp, vp],
        "gmrf_test_clock_probe_start": [i32, i32, i32, P(vp)],
        "gmrf_test_clock_probe_finish": [vp, vp, vp],
        "gmrf_test_mfma_f64_rate": [i32, P(dbl)],
        "gmrf_test_hbm_rate": [i32, i64, P(dbl)],
        "gmrf_test_microbench": [i32, vp, i32],
        "gmrf_test_symbolic_csc": [i64, i64, vp, vp, i32, vp],
    }
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = i32
    lib.gmrf_last_error.argtypes = []
    lib.gmrf_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def check(status: int, info: int = 0):
    if status == GMRF_OK:
        return
    msg = load().gmrf_last_error().decode(errors="replace")
    if status == ERR_NOT_SPD:
        raise NotPositiveDefinite(status, msg, info)
    raise GmrfError(status, msg, info)


def ptr(a) -> C.c_void_p:
    """Pointer of a NumPy array, a torch tensor (host or device) or a raw integer address."""
    if a is None:
        return C.c_void_p(0)
    if isinstance(a, int):
        return C.c_void_p(a)
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(a.ctypes.data)

"""ctypes binding of include/lbm_d2q9.h (liblbm_d2q9.so).

There is no CPU fallback: if the library is missing this module raises, and every device entry
point raises LbmError with the library's message on a non-zero status.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
# LBM_LIBRARY: another build of the same ABI in place of the shipped library (tests/experiments_suite.py runs on lib/variants/experiments.so)
LIB_PATH = os.environ.get("LBM_LIBRARY") or os.path.join(PKG, "lib", "liblbm_d2q9.so")

ABI_VERSION = 5
NSPEEDS = 9

FLAG_DEFAULT, FLAG_NT_STORES, FLAG_NO_NT_STORES, FLAG_KERNEL_LDS, FLAG_FORCE_HALO, FLAG_GRAPH, FLAG_ONE_STEP, FLAG_FAST_AVVELS, FLAG_EXACT_AVVELS = 0, 1, 2, 4, 8, 16, 32, 64, 128


class LbmError(RuntimeError):
    """A C-ABI call returned non-zero; str(e) is lbm_last_error()."""


class CParams(C.Structure):
    """struct lbm_params."""

    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("max_iters", C.c_int), ("reynolds_dim", C.c_int),
                ("density", C.c_float), ("accel", C.c_float), ("omega", C.c_float)]


class CLayout(C.Structure):
    """struct lbm_layout."""

    _fields_ = [("y0", C.c_int), ("ny_local", C.c_int), ("macro_k", C.c_int), ("ghost", C.c_int), ("group", C.c_int)]


class CTileLayout(C.Structure):
    """struct lbm_tile_layout."""

    _fields_ = [(n, C.c_int) for n in ("px", "py", "rx", "ry", "x0", "nx_local", "y0", "ny_local", "macro_k", "ghost", "group", "ghost_x", "ghost_y")]


_P = C.POINTER
_ctx = C.c_void_p
_SIGNATURES = {
    "lbm_abi_version": (C.c_int, []),
    "lbm_last_error": (C.c_char_p, []),
    "lbm_read_params": (C.c_int, [C.c_char_p, _P(CParams)]),
    "lbm_read_obstacles": (C.c_int, [C.c_char_p, C.c_int, C.c_int, _P(C.c_int), _P(C.c_int)]),
    "lbm_decompose": (C.c_int, [C.c_int, C.c_int, _P(C.c_int), _P(C.c_int)]),
    "lbm_plan_steps": (C.c_int, [C.c_int, C.c_int, C.c_int, _P(C.c_int), C.c_int]),
    "lbm_plan_group": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _P(C.c_int), C.c_int]),
    "lbm_create": (C.c_int, [_P(_ctx), _P(CParams), C.c_int, _P(C.c_int), C.c_int, C.c_int, C.c_int, C.c_uint]),
    "lbm_create_global": (C.c_int, [_P(_ctx), _P(CParams), C.c_int, _P(C.c_int), C.c_int, C.c_int, C.c_int, C.c_uint]),
    "lbm_rank_layout": (C.c_int, [_P(CParams), C.c_int, C.c_int, C.c_uint, _P(CLayout)]),
    "lbm_create_rank": (C.c_int, [_P(_ctx), _P(CParams), C.c_int, _P(C.c_int), C.c_int, C.c_int, C.c_int, C.c_uint]),
    "lbm_decompose_columns": (C.c_int, [C.c_int, C.c_int, _P(C.c_int), _P(C.c_int)]),
    "lbm_tile_layout_of": (C.c_int, [_P(CParams), C.c_int, C.c_int, C.c_int, C.c_uint, _P(CTileLayout)]),
    "lbm_choose_rank_grid": (C.c_int, [_P(CParams), C.c_int, C.c_uint, _P(C.c_int), _P(C.c_int)]),
    "lbm_create_tile": (C.c_int, [_P(_ctx), _P(CParams), C.c_int, _P(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint]),
    "lbm_tile_info": (C.c_int, [_ctx, _P(CTileLayout)]),
    "lbm_destroy": (C.c_int, [_ctx]),
    "lbm_run": (C.c_int, [_ctx, C.c_int, _P(C.c_float)]),
    "lbm_get_cells": (C.c_int, [_ctx, _P(C.c_float)]),
    "lbm_set_cells": (C.c_int, [_ctx, _P(C.c_float)]),
    "lbm_get_observables": (C.c_int, [_ctx, _P(C.c_float)]),
    "lbm_state_checksum": (C.c_int, [_ctx, C.c_int, C.c_int, _P(C.c_ulonglong)]),
    "lbm_av_velocity_sum": (C.c_int, [_ctx, _P(C.c_double)]),
    "lbm_halo_floats": (C.c_size_t, [_ctx]),
    "lbm_halo_send_ptr": (C.c_void_p, [_ctx, C.c_int]),
    "lbm_halo_recv_ptr": (C.c_void_p, [_ctx, C.c_int]),
    "lbm_bind_halo_buffers": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lbm_step_prepare": (C.c_int, [_ctx, C.c_int, C.c_void_p]),
    "lbm_step_interior": (C.c_int, [_ctx, C.c_void_p]),
    "lbm_step_boundary": (C.c_int, [_ctx, C.c_void_p]),
    "lbm_step_finish": (C.c_int, [_ctx, C.c_void_p]),
    "lbm_macro_steps": (C.c_int, [_ctx]),
    "lbm_macro_next_steps": (C.c_int, [_ctx]),
    "lbm_macro_next_launches": (C.c_int, [_ctx]),
    "lbm_macro_all": (C.c_int, [_ctx, C.c_void_p]),
    "lbm_macro_halo_floats": (C.c_size_t, [_ctx]),
    "lbm_macro_send_ptr": (C.c_void_p, [_ctx, C.c_int, C.c_int]),
    "lbm_macro_recv_ptr": (C.c_void_p, [_ctx, C.c_int, C.c_int]),
    "lbm_macro_pack_floats": (C.c_size_t, [_ctx]),
    "lbm_macro_pack_ptr": (C.c_void_p, [_ctx, C.c_int, C.c_int]),
    "lbm_macro_pack": (C.c_int, [_ctx, C.c_void_p]),
    "lbm_macro_unpack": (C.c_int, [_ctx, C.c_void_p]),
    "lbm_macro_prepare": (C.c_int, [_ctx, C.c_int, C.c_void_p]),
    "lbm_macro_interior": (C.c_int, [_ctx, C.c_void_p]),
    "lbm_macro_edge": (C.c_int, [_ctx, C.c_void_p]),
    "lbm_macro_finish": (C.c_int, [_ctx, C.c_void_p]),
    "lbm_macro_exchange_local": (C.c_int, [_ctx, _ctx, C.c_int, C.c_void_p]),
    "lbm_step_fold": (C.c_int, [_ctx, C.c_void_p]),
    "lbm_step_collect": (C.c_int, [_ctx, C.c_void_p, _P(C.c_double), C.c_int]),
    "lbm_step_sums_device_ptr": (C.c_void_p, [_ctx]),
    "lbm_last_run_kernel_ms": (C.c_int, [_ctx, _P(C.c_double), _P(C.c_int)]),
    "lbm_set_profile": (C.c_int, [_ctx, C.c_int]),
    "lbm_launch_profile": (C.c_int, [_ctx, C.c_int, _P(C.c_int), _P(C.c_double), _P(C.c_int)]),
    "lbm_device": (C.c_int, [_ctx]),
    "lbm_stream": (C.c_void_p, [_ctx]),
    "lbm_describe": (C.c_int, [_ctx, C.c_char_p, C.c_size_t, _P(C.c_longlong), _P(C.c_longlong)]),
    "lbm_av_velocity_host": (C.c_float, [_P(CParams), _P(C.c_float), _P(C.c_int), C.c_int]),
    "lbm_av_velocity_obs": (C.c_float, [_P(CParams), _P(C.c_float), _P(C.c_int), C.c_int]),
    "lbm_reynolds": (C.c_float, [_P(CParams), C.c_float]),
    "lbm_write_final_state": (C.c_int, [C.c_char_p, _P(CParams), _P(C.c_float), _P(C.c_int), C.c_int, C.c_int, C.c_int]),
    "lbm_write_final_state_obs": (C.c_int, [C.c_char_p, _P(CParams), _P(C.c_float), _P(C.c_int), C.c_int, C.c_int, C.c_int]),
    "lbm_write_av_vels": (C.c_int, [C.c_char_p, _P(C.c_float), C.c_int]),
}
EXPORTS = tuple(_SIGNATURES)

# include/lbm_d2q9_p2p.h (same library)
P2P_HANDLE_BYTES = 512
_P2P_SIGNATURES = {
    "lbm_p2p_create": (C.c_int, [_P(C.c_void_p), _ctx, C.c_int, C.c_int]),
    "lbm_p2p_handle": (C.c_int, [C.c_void_p, C.c_char_p]),
    "lbm_p2p_connect": (C.c_int, [C.c_void_p, C.c_char_p]),
    "lbm_p2p_disconnect": (C.c_int, [C.c_void_p]),
    "lbm_p2p_destroy": (C.c_int, [C.c_void_p]),
    "lbm_p2p_run": (C.c_int, [C.c_void_p, C.c_int, _P(C.c_double)]),
    "lbm_p2p_describe": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "lbm_p2p_set_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "lbm_p2p_phases": (C.c_int, [C.c_void_p, _P(C.c_double)]),
    "lbm_p2p_phase_name": (C.c_char_p, [C.c_int]),
}
P2P_PHASES = 16
P2P_EXPORTS = tuple(_P2P_SIGNATURES)

_RCCL_SIGNATURES = {
    "lbm_comm_unique_id": (C.c_int, [C.c_char_p]),
    "lbm_comm_create": (C.c_int, [_P(C.c_void_p), _ctx, C.c_char_p, C.c_int, C.c_int]),
    "lbm_comm_destroy": (C.c_int, [C.c_void_p]),
    "lbm_comm_run": (C.c_int, [C.c_void_p, C.c_int, _P(C.c_double)]),
    "lbm_comm_nranks": (C.c_int, [C.c_void_p]),
    "lbm_comm_set_step_allreduce": (C.c_int, [C.c_void_p, C.c_int]),
}
RCCL_EXPORTS = tuple(_RCCL_SIGNATURES)
COMM_ID_BYTES = 128
LIB_RCCL_PATH = os.path.join(PKG, "lib", "liblbm_d2q9_rccl.so")

_lib = None
_lib_rccl = None


def load_rccl_library() -> C.CDLL:
    """liblbm_d2q9_rccl.so (include/lbm_d2q9_rccl.h).  torch is imported first so that the process
    holds ONE RCCL and ONE HIP runtime (torch's bundled ones carry the same SONAMEs)."""
    global _lib_rccl
    if _lib_rccl is None:
        load_library()
        if not os.path.exists(LIB_RCCL_PATH):
            raise RuntimeError(f"{LIB_RCCL_PATH} is missing: run build()")
        lib = C.CDLL(LIB_RCCL_PATH)
        for name, (res, args) in _RCCL_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib_rccl = lib
    return _lib_rccl


def load_library() -> C.CDLL:
    """Load liblbm_d2q9.so from the package tree and type every export.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the HIP extension is the only implementation of the timestep path)")
    try:
        import torch  # noqa: F401  -- first, so its bundled HIP runtime is the only one in the process
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in {**_SIGNATURES, **_P2P_SIGNATURES}.items():
        fn = getattr(lib, name)           # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.lbm_abi_version() != ABI_VERSION:
        raise RuntimeError(f"liblbm_d2q9.so ABI {lib.lbm_abi_version()} != binding {ABI_VERSION}: rebuild")
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != 0:
        raise LbmError(load_library().lbm_last_error().decode(errors="replace"))


def as_float_ptr(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(_P(C.c_float))


def as_int_ptr(a: np.ndarray):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(_P(C.c_int))


def as_double_ptr(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(_P(C.c_double))

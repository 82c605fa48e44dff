"""Loader for the CPU oracle (TEST INFRASTRUCTURE). Importable only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from ryujin_amd import capi  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import build_oracle as _build_oracle  # noqa: E402

EXCHANGE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_int)

_lib = None


def load(path: str | None = None):
    global _lib
    if _lib is not None and path is None:
        return _lib
    if path is None:
        path = os.environ.get("RYUJIN_ORACLE_LIB", _build_oracle.ORACLE_SO)  # (the sanitizer builds: scripts/sanitizer_run.sh)
        if not os.path.exists(path):
            _build_oracle.build_oracle()
    lib = C.CDLL(path)
    capi._declare_module_api(lib, "ryujin_oracle_")
    dp = capi.c_double_p
    lib.ryujin_oracle_set_exchange.argtypes = [C.c_void_p, EXCHANGE_FN, C.c_void_p]
    lib.ryujin_oracle_set_exchange.restype = None
    lib.ryujin_oracle_set_expensive_bounds_check.argtypes = [C.c_void_p, C.c_int]
    lib.ryujin_oracle_set_expensive_bounds_check.restype = None
    lib.ryujin_oracle_set_flush_denormals.argtypes = [C.c_int]
    lib.ryujin_oracle_set_flush_denormals.restype = None
    lib.ryujin_oracle_euler_riemann.argtypes = [C.POINTER(capi.Params), dp, dp, dp, dp, C.c_int]
    lib.ryujin_oracle_euler_lambda_max.argtypes = [C.POINTER(capi.Params), dp, dp, dp]
    lib.ryujin_oracle_euler_lambda_max.restype = C.c_double
    lib.ryujin_oracle_euler_dij_batch.argtypes = [C.POINTER(capi.Params), C.c_size_t, dp, dp, dp, dp]
    lib.ryujin_oracle_euler_limit_1d.argtypes = [C.POINTER(capi.Params), C.c_int, dp, dp, dp, dp, dp, C.c_int]
    lib.ryujin_oracle_euler_limit.argtypes = [C.POINTER(capi.Params), dp, dp, dp, dp, capi.c_int_p]
    lib.ryujin_oracle_euler_limit_trace.argtypes = [C.POINTER(capi.Params), dp, dp, dp, dp]
    lib.ryujin_oracle_euler_view.argtypes = [C.POINTER(capi.Params), dp, dp]
    lib.ryujin_oracle_euler_apply_bc.argtypes = [C.POINTER(capi.Params), C.c_int, dp, dp, dp, dp]
    lib.ryujin_oracle_sw_dij_batch.argtypes = [C.POINTER(capi.Params), C.c_size_t, dp, dp, dp, dp]
    lib.ryujin_oracle_sw_riemann.argtypes = [C.POINTER(capi.Params), dp, dp, dp]
    lib.ryujin_oracle_import_csr.argtypes = [C.POINTER(capi.Offline), capi.c_u64_p, capi.c_u32_p,
                                             capi.c_u64_p, dp, C.c_uint32, dp]
    pp = C.POINTER(capi.Params)
    lib.ryujin_oracle_aeos_riemann.argtypes = [pp, dp, dp, dp, dp]
    lib.ryujin_oracle_aeos_lambda_max.argtypes = [pp, dp, dp, dp]
    lib.ryujin_oracle_aeos_lambda_max.restype = C.c_double
    lib.ryujin_oracle_aeos_limit.argtypes = [pp, C.c_int, dp, dp, dp, dp, capi.c_int_p, dp, C.c_int]
    lib.ryujin_oracle_aeos_view.argtypes = [pp, dp, C.c_double, dp]
    lib.ryujin_oracle_aeos_eos.argtypes = [pp, C.c_double, C.c_double, C.c_double, dp]
    lib.ryujin_oracle_scalar_riemann.argtypes = [pp, C.c_double, C.c_double, dp, dp, dp]
    lib.ryujin_oracle_scalar_flux.argtypes = [pp, C.c_double, dp]
    lib.ryujin_oracle_scalar_limit.argtypes = [pp, C.c_int, dp, C.c_double, C.c_double, dp, capi.c_int_p]
    if path == _build_oracle.ORACLE_SO:
        _lib = lib
    return lib


def lib():
    return load()


def backend(path: str | None = None):
    """(lib, prefix) pair accepted by ryujin_amd.HyperbolicModule(backend=...)."""
    return (load(path), "ryujin_oracle_")


def default_params(equation=capi.EQ_EULER, dim=2) -> capi.Params:
    p = capi.Params()
    load().ryujin_oracle_default_params(C.byref(p), equation, dim)
    return p

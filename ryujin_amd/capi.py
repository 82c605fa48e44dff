"""ctypes mirror of include/ryujin_hip.h and include/ryujin_synth.h.

Python is plumbing here (tests, bench.py); the product is libryujin_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _build

c_double_p = C.POINTER(C.c_double)
c_u32_p = C.POINTER(C.c_uint32)
c_u64_p = C.POINTER(C.c_uint64)
c_u8_p = C.POINTER(C.c_uint8)
c_int_p = C.POINTER(C.c_int)

RYUJIN_OK, RYUJIN_WARN, RYUJIN_RESTART = 0, 1, 2
RYUJIN_ERR_TAU, RYUJIN_ERR_ARG, RYUJIN_ERR_HIP, RYUJIN_ERR_COMM, RYUJIN_ERR_UNSUPPORTED = -1, -2, -3, -4, -5
EQ_EULER, EQ_SHALLOW_WATER, EQ_EULER_AEOS, EQ_SCALAR_CONSERVATION = 0, 1, 2, 3
FLUX_BURGERS, FLUX_KPP, FLUX_POLYNOMIAL = 0, 1, 2
EOS_POLYTROPIC_GAS, EOS_NOBLE_ABEL_STIFFENED_GAS, EOS_VAN_DER_WAALS, EOS_JONES_WILKINS_LEE = 0, 1, 2, 3
BC_DO_NOTHING, BC_PERIODIC, BC_SLIP, BC_NO_SLIP, BC_DIRICHLET, BC_DYNAMIC, BC_DIRICHLET_MOMENTUM = range(7)
IDV_WARN, IDV_RAISE_EXCEPTION = 0, 1
CUT_NONE, CUT_BOX, CUT_CYLINDER = 0, 1, 2
SCHEME_SSPRK_22, SCHEME_SSPRK_33, SCHEME_ERK_11, SCHEME_ERK_22, SCHEME_ERK_33, SCHEME_ERK_43, SCHEME_ERK_54 = range(7)
CFL_RECOVERY_NONE, CFL_RECOVERY_BANG_BANG = 0, 1
UNIQUE_ID_BYTES = 128
DEBUG_EULER_RIEMANN, DEBUG_EULER_LIMIT_1D, DEBUG_SW_RIEMANN, DEBUG_EULER_DIJ_2D, DEBUG_EULER_DIJ_3D = range(5)
DEBUG_EULER_DIJ_RECORDS_2D, DEBUG_EULER_DIJ_RECORDS_3D = 5, 6
DEBUG_SW_DIJ_2D, DEBUG_SW_DIJ_RECORDS_2D = 7, 8
DEBUG_EULER_RIEMANN_RECORDS, DEBUG_SW_RIEMANN_RECORDS = 9, 10
DEBUG_AEOS_RIEMANN, DEBUG_AEOS_LIMIT_1D = 11, 12
DEBUG_AEOS_DIJ_2D, DEBUG_AEOS_DIJ_RECORDS_2D = 13, 14
DEBUG_EULER_LIMIT_CHECKED_1D, DEBUG_EULER_LIMIT_2D = 15, 16


DIRICHLET_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_double, c_double_p)   # ryujin_hip_dirichlet_fn


class Params(C.Structure):
    _fields_ = [
        ("equation", C.c_int), ("dim", C.c_int),
        ("gamma", C.c_double), ("reference_density", C.c_double),
        ("vacuum_state_relaxation_small", C.c_double), ("vacuum_state_relaxation_large", C.c_double),
        ("gravity", C.c_double), ("manning_friction_coefficient", C.c_double),
        ("reference_water_depth", C.c_double), ("dry_state_relaxation_factor", C.c_double),
        ("dry_state_relaxation_small", C.c_double), ("dry_state_relaxation_large", C.c_double),
        ("cfl", C.c_double), ("id_violation_strategy", C.c_int),
        ("indicator_evc_factor", C.c_double),
        ("limiter_iterations", C.c_int), ("limiter_newton_tolerance", C.c_double),
        ("limiter_newton_max_iterations", C.c_int), ("limiter_relaxation_factor", C.c_double),
        ("limiter_limit_on_kinetic_energy", C.c_int), ("limiter_limit_on_square_velocity", C.c_int),
        ("riemann_newton_max_iterations", C.c_int), ("riemann_newton_tolerance", C.c_double),
        ("eos", C.c_int), ("compute_strict_bounds", C.c_int),
        ("eos_covolume_b", C.c_double), ("eos_q", C.c_double), ("eos_pinf", C.c_double),
        ("eos_vdw_a", C.c_double), ("eos_gas_constant_R", C.c_double),
        ("jwl_A", C.c_double), ("jwl_B", C.c_double), ("jwl_R1", C.c_double), ("jwl_R2", C.c_double),
        ("jwl_omega", C.c_double), ("jwl_rho_0", C.c_double), ("jwl_q_0", C.c_double),
        ("jwl_cv", C.c_double),
        ("sc_flux", C.c_int), ("sc_flux_polynomial", (C.c_double * 4) * 3),
        ("sc_derivative_approximation_delta", C.c_double), ("sc_use_greedy_wavespeed", C.c_int),
        ("sc_use_averaged_entropy", C.c_int), ("sc_random_entropies", C.c_int),
        # run-time switches of the library (no ParameterAcceptor counterpart; 0 = default)
        ("system_scope_events", C.c_int), ("debug_join_exchanges", C.c_int),
        ("debug_bc_fold_max_slices", C.c_int), ("debug_no_small_mesh_split", C.c_int),
        ("debug_pij_storage", C.c_int), ("debug_expensive_bounds_check", C.c_int),
        ("debug_tile_map", C.c_int), ("debug_band_stride", C.c_int),
        ("debug_xcd_chunk", C.c_int),
    ]


class Offline(C.Structure):
    _fields_ = [
        ("n_export", C.c_uint32), ("n_internal", C.c_uint32), ("n_owned", C.c_uint32),
        ("n_relevant", C.c_uint32), ("simd_length", C.c_uint32),
        ("row_starts", c_u64_p), ("columns", c_u32_p), ("cij", c_double_p), ("mij", c_double_p),
        ("mi", c_double_p), ("mi_inv", c_double_p), ("measure_of_omega", C.c_double),
        ("n_bdry", C.c_uint32), ("b_i", c_u32_p), ("b_normal", c_double_p), ("b_id", c_u8_p),
        ("n_pairs", C.c_uint32), ("p_i", c_u32_p), ("p_col", c_u32_p), ("p_j", c_u32_p),
        ("initial_precomputed", c_double_p),
        ("n_nbr", C.c_int), ("nbr_rank", c_int_p), ("send_off", c_u32_p), ("send_idx", c_u32_p),
        ("recv_off", c_u32_p), ("row_send_off", c_u32_p), ("row_send_row", c_u32_p),
        ("row_send_col", c_u32_p),
        ("discontinuous_ansatz", C.c_int), ("incidence", c_double_p), ("mass_matrix_inverse", c_double_p),
    ]


class SynthSpec(C.Structure):
    _fields_ = [
        ("dim", C.c_int), ("n_cells", C.c_uint32 * 3), ("lower", C.c_double * 3),
        ("upper", C.c_double * 3), ("bc", C.c_int * 6), ("cut_kind", C.c_int),
        ("cut_lo", C.c_double * 3), ("cut_hi", C.c_double * 3), ("cyl_center", C.c_double * 2),
        ("cyl_radius", C.c_double), ("cut_bc", C.c_int), ("n_ranks", C.c_int), ("rank", C.c_int),
    ]


def as_ptr(a: np.ndarray, typ):
    return a.ctypes.data_as(typ)


def np_from_ptr(ptr, n: int, dtype) -> np.ndarray:
    """Copy n items from a ctypes pointer into a fresh numpy array."""
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


# --------------------------------------------------------------------------- synth

_synth = None


def load_synth():
    global _synth
    if _synth is None:
        path = os.environ.get("RYUJIN_SYNTH_LIB", _build.SYNTH_SO)  # (the sanitizer builds: scripts/sanitizer_run.sh)
        if not os.path.exists(path):
            _build.build_synth()
        lib = C.CDLL(path)
        lib.ryujin_synth_build.restype = C.c_void_p
        lib.ryujin_synth_build.argtypes = [C.POINTER(SynthSpec)]
        lib.ryujin_synth_free.argtypes = [C.c_void_p]
        lib.ryujin_synth_last_error.restype = C.c_char_p
        lib.ryujin_synth_offline.restype = C.POINTER(Offline)
        lib.ryujin_synth_offline.argtypes = [C.c_void_p]
        lib.ryujin_synth_nnz.restype = C.c_uint64
        lib.ryujin_synth_nnz.argtypes = [C.c_void_p]
        lib.ryujin_synth_n_global.restype = C.c_uint64
        lib.ryujin_synth_n_global.argtypes = [C.c_void_p]
        lib.ryujin_synth_positions.restype = c_double_p
        lib.ryujin_synth_positions.argtypes = [C.c_void_p]
        lib.ryujin_synth_global_ids.restype = c_u64_p
        lib.ryujin_synth_global_ids.argtypes = [C.c_void_p]
        lib.ryujin_synth_bdry_positions.restype = c_double_p
        lib.ryujin_synth_bdry_positions.argtypes = [C.c_void_p]
        lib.ryujin_synth_ghost_row_send_entries.restype = C.c_size_t
        lib.ryujin_synth_ghost_row_send_entries.argtypes = [c_u64_p, c_u32_p, c_u32_p, C.c_size_t, C.c_uint32,
                                                            C.c_uint32, c_u32_p, c_u32_p]
        # OfflineData dumps (include/ryujin_offline_io.h)
        lib.ryujin_offline_write.restype = C.c_int
        lib.ryujin_offline_write.argtypes = [C.c_char_p, C.POINTER(Offline), C.c_int, C.c_int, c_double_p,
                                             c_double_p]
        lib.ryujin_offline_read.restype = C.c_void_p
        lib.ryujin_offline_read.argtypes = [C.c_char_p]
        lib.ryujin_offline_file_free.argtypes = [C.c_void_p]
        lib.ryujin_offline_io_last_error.restype = C.c_char_p
        lib.ryujin_offline_file_view.restype = C.POINTER(Offline)
        lib.ryujin_offline_file_view.argtypes = [C.c_void_p]
        lib.ryujin_offline_file_dim.argtypes = [C.c_void_p]
        lib.ryujin_offline_file_n_initial_precomputed.argtypes = [C.c_void_p]
        lib.ryujin_offline_file_nnz.restype = C.c_uint64
        lib.ryujin_offline_file_nnz.argtypes = [C.c_void_p]
        lib.ryujin_offline_file_positions.restype = c_double_p
        lib.ryujin_offline_file_positions.argtypes = [C.c_void_p]
        lib.ryujin_offline_file_b_positions.restype = c_double_p
        lib.ryujin_offline_file_b_positions.argtypes = [C.c_void_p]
        _synth = lib
    return _synth


# --------------------------------------------------------------------------- hip

HIP_SYMBOLS = [
    "ryujin_hip_comm_unique_id", "ryujin_hip_comm_init", "ryujin_hip_comm_init_local",
    "ryujin_hip_comm_init_loopback",
    "ryujin_hip_comm_destroy", "ryujin_hip_comm_info", "ryujin_hip_exchange_info", "ryujin_hip_device_count",
    "ryujin_hip_time_step_fn", "ryujin_hip_debug_addresses",
    "ryujin_hip_host_register", "ryujin_hip_host_unregister", "ryujin_hip_state_download_owned",
    "ryujin_hip_state_download_prepared", "ryujin_hip_layout_info", "ryujin_hip_chain_info", "ryujin_hip_tile_statistics",
    "ryujin_hip_deferred_slices",
    "ryujin_hip_default_params", "ryujin_hip_create", "ryujin_hip_destroy",
    "ryujin_hip_state_alloc", "ryujin_hip_state_free", "ryujin_hip_state_upload",
    "ryujin_hip_state_download", "ryujin_hip_state_download_precomputed", "ryujin_hip_state_integrals",
    "ryujin_hip_prepare_state_vector", "ryujin_hip_step", "ryujin_hip_sadd", "ryujin_hip_time_step", "ryujin_hip_time_step_n",
    "ryujin_hip_get_timers_accum",
    "ryujin_hip_set_cfl", "ryujin_hip_get_cfl", "ryujin_hip_set_id_violation_strategy",
    "ryujin_hip_get_alpha", "ryujin_hip_get_counters", "ryujin_hip_limiter_statistics", "ryujin_hip_debug_fetch",
    "ryujin_hip_set_timers", "ryujin_hip_get_timers", "ryujin_hip_synchronize",
    "ryujin_hip_event_record", "ryujin_hip_event_elapsed_ms", "ryujin_hip_last_error",
    "ryujin_hip_version", "ryujin_hip_debug_layout", "ryujin_hip_debug_pow", "ryujin_hip_debug_function", "ryujin_hip_debug_rk_outcome",
]


def _declare_module_api(lib, prefix: str):
    """Declare argtypes of the HyperbolicModule C ABI on `lib` (same signatures for the HIP
    library, prefix 'ryujin_hip_', and the CPU oracle, prefix 'ryujin_oracle_')."""
    p = lambda n: getattr(lib, prefix + n)  # noqa: E731
    vp = C.c_void_p
    p("default_params").argtypes = [C.POINTER(Params), C.c_int, C.c_int]
    p("default_params").restype = None
    p("create").argtypes = [C.POINTER(vp), C.POINTER(Offline), C.POINTER(Params), vp, C.c_int]
    p("destroy").argtypes = [vp]
    p("destroy").restype = None
    p("state_alloc").argtypes = [vp, c_int_p]
    p("state_free").argtypes = [vp, C.c_int]
    p("state_upload").argtypes = [vp, C.c_int, c_double_p]
    p("state_download").argtypes = [vp, C.c_int, c_double_p]
    p("state_download_precomputed").argtypes = [vp, C.c_int, c_double_p]
    if hasattr(lib, prefix + "state_integrals"):  # device library only
        p("state_integrals").argtypes = [vp, C.c_int, c_double_p]
    p("prepare_state_vector").argtypes = [vp, C.c_int, C.c_double, c_double_p]
    p("step").argtypes = [vp, C.c_int, C.c_int, c_int_p, c_double_p, C.c_int, C.c_double,
                          C.c_double, c_double_p]
    p("sadd").argtypes = [vp, C.c_int, C.c_double, C.c_double, C.c_int]
    p("set_cfl").argtypes = [vp, C.c_double]
    p("get_cfl").argtypes = [vp, c_double_p]
    p("set_id_violation_strategy").argtypes = [vp, C.c_int]
    p("get_alpha").argtypes = [vp, c_double_p]
    p("get_counters").argtypes = [vp, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    p("debug_fetch").argtypes = [vp, C.c_int, c_double_p, C.c_size_t]
    p("last_error").restype = C.c_char_p


_hip = None


def load_hip():
    """Load the HIP library. Fails loudly if it is missing: there is no CPU fallback."""
    global _hip
    if _hip is None:
        path = os.environ.get("RYUJIN_HIP_LIB", _build.HIP_SO)  # A/B variants of the same library
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with `python -m ryujin_amd._build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
        lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        _declare_module_api(lib, "ryujin_hip_")
        vp = C.c_void_p
        lib.ryujin_hip_comm_unique_id.argtypes = [C.c_char_p]
        lib.ryujin_hip_comm_init.argtypes = [C.POINTER(vp), C.c_char_p, C.c_int, C.c_int, C.c_int]
        lib.ryujin_hip_comm_init_local.argtypes = [C.POINTER(vp), C.c_int, C.c_int]
        lib.ryujin_hip_comm_init_loopback.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int]
        lib.ryujin_hip_comm_destroy.argtypes = [vp]
        lib.ryujin_hip_comm_destroy.restype = None
        lib.ryujin_hip_comm_info.argtypes = [vp, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p]
        lib.ryujin_hip_exchange_info.argtypes = [vp, c_int_p, c_int_p, C.c_int, C.POINTER(C.c_ulonglong),
                                                 C.POINTER(C.c_ulonglong)]
        lib.ryujin_hip_set_timers.argtypes = [vp, C.c_int]
        lib.ryujin_hip_get_timers.argtypes = [vp, c_double_p]
        lib.ryujin_hip_get_timers_accum.argtypes = [vp, c_double_p, C.POINTER(C.c_uint), C.c_int]
        lib.ryujin_hip_time_step.argtypes = [vp, C.c_int, C.c_int, c_int_p, c_double_p, C.c_double, C.c_int,
                                             C.c_double, C.c_double, c_double_p]
        lib.ryujin_hip_time_step_n.argtypes = [vp, C.c_int, C.c_int, C.c_int, c_int_p, c_double_p, C.c_double,
                                               C.c_int, C.c_double, C.c_double, c_double_p]
        lib.ryujin_hip_time_step_fn.argtypes = [vp, C.c_int, C.c_int, C.c_int, c_int_p, C.c_double, DIRICHLET_FN,
                                                C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_double, c_double_p]
        lib.ryujin_hip_device_count.argtypes = [c_int_p]
        lib.ryujin_hip_layout_info.argtypes = [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
        lib.ryujin_hip_chain_info.argtypes = [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
        lib.ryujin_hip_tile_statistics.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        lib.ryujin_hip_deferred_slices.argtypes = [vp, C.POINTER(C.c_uint)]
        lib.ryujin_hip_host_register.argtypes = [vp, C.c_void_p, C.c_size_t]
        lib.ryujin_hip_host_unregister.argtypes = [vp, C.c_void_p]
        lib.ryujin_hip_state_download_owned.argtypes = [vp, C.c_int, c_double_p]
        lib.ryujin_hip_state_download_prepared.argtypes = [vp, C.c_int, c_double_p]
        lib.ryujin_hip_debug_addresses.argtypes = [vp, c_u64_p]
        lib.ryujin_hip_synchronize.argtypes = [vp]
        lib.ryujin_hip_event_record.argtypes = [vp, C.c_int]
        lib.ryujin_hip_event_elapsed_ms.argtypes = [vp, c_double_p]
        lib.ryujin_hip_version.restype = C.c_char_p
        lib.ryujin_hip_debug_layout.argtypes = [C.POINTER(Offline), c_u64_p, c_u32_p, c_u64_p,
                                                c_double_p, C.c_uint32, c_double_p]
        lib.ryujin_hip_debug_pow.argtypes = [C.c_int, c_double_p, c_double_p, c_double_p, C.c_size_t]
        lib.ryujin_hip_debug_rk_outcome.argtypes = [C.c_int, C.c_int, C.c_int]
        lib.ryujin_hip_debug_function.argtypes = [C.c_int, C.POINTER(Params), C.c_int, c_double_p, c_double_p,
                                                  C.c_size_t]
        _hip = lib
    return _hip

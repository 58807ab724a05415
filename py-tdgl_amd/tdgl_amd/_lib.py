"""ctypes binding of libtdgl_hip.so (the C ABI declared in include/tdgl_hip.h).

There is deliberately no fallback: if the library is missing or no MI355X is visible, the
product path raises.  (The NumPy oracle under /oracle is test infrastructure only and is
never imported from here.)
"""

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (TDGL_HIP_LIB: load another build of the library, e.g. a variant under measurement)
LIB_PATH = os.environ.get("TDGL_HIP_LIB") or os.path.join(_HERE, "lib", "libtdgl_hip.so")

TDGL_OK = 0
TDGL_ERR_HIP = 1
TDGL_ERR_ARG = 2
TDGL_ERR_PSI_RETRIES = 3
TDGL_ERR_PCG = 4
TDGL_ERR_NOT_READY = 5
TDGL_ERR_SCREENING = 6

c_i32p = C.POINTER(C.c_int32)
c_f64p = C.POINTER(C.c_double)


class MeshDesc(C.Structure):
    _fields_ = [
        ("n_sites", C.c_int64),
        ("n_edges", C.c_int64),
        ("n_boundary_edges", C.c_int64),
        ("edges", c_i32p),
        ("areas", c_f64p),
        ("edge_lengths", c_f64p),
        ("dual_edge_lengths", c_f64p),
        ("directions", c_f64p),
        ("boundary_edge_indices", c_i32p),
        ("fixed_sites", c_i32p),
        ("n_fixed", C.c_int64),
        ("fix_psi", C.c_int32),
        ("site_perm", c_i32p),
        ("u", C.c_double),
        ("gamma", C.c_double),
        ("n_owned", C.c_int64),
    ]


class AmgLevel(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("n_coarse", C.c_int64),
        ("A_indptr", c_i32p),
        ("A_indices", c_i32p),
        ("A_data", c_f64p),
        ("dinv", c_f64p),
        ("rho", C.c_double),
        ("P_indptr", c_i32p),
        ("P_indices", c_i32p),
        ("P_data", c_f64p),
        ("R_indptr", c_i32p),
        ("R_indices", c_i32p),
        ("R_data", c_f64p),
        ("n_cols", C.c_int64),
        ("a_rows", C.c_int64),
        ("p_rows", C.c_int64),
        ("explicit_only", C.c_int32),
    ]


class DeepHaloPlan(C.Structure):
    _fields_ = [
        ("n_ext", C.c_int64),
        ("n_neighbors", C.c_int32),
        ("neighbor_ranks", c_i32p),
        ("send_ptr", c_i32p),
        ("send_idx", c_i32p),
        ("recv_ptr", c_i32p),
        ("recv_idx", c_i32p),
        ("l1_interior", C.c_int64),
    ]


class HaloPlan(C.Structure):
    _fields_ = [
        ("rank", C.c_int32),
        ("world", C.c_int32),
        ("n_global", C.c_int64),
        ("n_neighbors", C.c_int32),
        ("neighbor_ranks", c_i32p),
        ("send_ptr", c_i32p),
        ("send_idx", c_i32p),
        ("recv_ptr", c_i32p),
    ]


c_i64p = C.POINTER(C.c_int64)
HALO_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, c_f64p, c_i64p, c_f64p, c_i64p, C.c_int32, c_i32p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, c_f64p, C.c_int64, C.c_int32)


class Controller(C.Structure):
    _fields_ = [
        ("dt_init", C.c_double),
        ("dt_max", C.c_double),
        ("adaptive", C.c_int32),
        ("adaptive_window", C.c_int32),
        ("max_solve_retries", C.c_int32),
        ("adaptive_time_step_multiplier", C.c_double),
    ]


class PoissonOptions(C.Structure):
    _fields_ = [
        ("rtol", C.c_double),
        ("max_iter", C.c_int32),
        ("nu", C.c_int32),
        ("check_every", C.c_int32),
        ("edge_currents_every_step", C.c_int32),
        ("smoother", C.c_int32),
        ("cheb_lo", C.c_double),
        ("extrapolate", C.c_int32),
        ("nu_fine", C.c_int32),
        ("precond_fp32", C.c_int32),
        ("guess_window", C.c_int32),
        ("flexible_cg", C.c_int32),
    ]


class CollapsedTail(C.Structure):
    _fields_ = [
        ("level", C.c_int32),
        ("mode", C.c_int32),
        ("G", c_f64p),
        ("g_rows", C.c_int64),
        ("W_indptr", c_i32p),
        ("W_indices", c_i32p),
        ("W_data", c_f64p),
        ("V", c_f64p),
        ("v_cols", C.c_int64),
        ("nu", C.c_int32),
        ("smoother", C.c_int32),
        ("cheb_lo", C.c_double),
    ]


class Substructure(C.Structure):
    _fields_ = [
        ("n_interior", C.c_int64),
        ("n_sep", C.c_int64),
        ("n_parts", C.c_int32),
        ("part_ptr", c_i32p),
        ("seg_ptr", c_i32p),
        ("seg_val", C.POINTER(C.c_int64)),
        ("seg_x", c_i32p),
        ("seg_len", c_i32p),
        ("vals", c_f64p),
        ("n_vals", C.c_int64),
        ("sep_ptr", c_i32p),
        ("sep_idx", c_i32p),
        ("e_off", C.POINTER(C.c_int64)),
        ("e_vals", c_f64p),
        ("n_e", C.c_int64),
        ("u", c_f64p),
        ("schur", c_f64p),
    ]


class SchurPiece(C.Structure):
    """`tdgl_schur_piece` (include/tdgl_hip.h): one rank's share of the rank-level dissection."""

    _fields_ = [
        ("n_interior", C.c_int64),
        ("n_gamma", C.c_int64),
        ("n_gamma_owned", C.c_int64),
        ("interior", c_i32p),
        ("gamma_owned_local", c_i32p),
        ("gamma_owned_gid", c_i32p),
        ("gi_indptr", c_i32p),
        ("gi_indices", c_i32p),
        ("gi_data", c_f64p),
        ("ig_indptr", c_i32p),
        ("ig_indices", c_i32p),
        ("ig_data", c_f64p),
    ]


class SubstructurePlan(C.Structure):
    _fields_ = [
        ("n_interior", C.c_int64),
        ("n_sep", C.c_int64),
        ("n_parts", C.c_int32),
        ("part_ptr", c_i32p),
        ("sep_ptr", c_i32p),
        ("sep_idx", c_i32p),
        ("ent_ptr", c_i32p),
        ("ent_row", c_i32p),
        ("ent_val", c_f64p),
        ("node_ptr", c_i32p),
        ("node_pair", c_i32p),
        ("ass_indptr", c_i32p),
        ("ass_indices", c_i32p),
        ("ass_data", c_f64p),
        ("seg_ptr", c_i32p),
        ("seg_val", C.POINTER(C.c_int64)),
        ("seg_x", c_i32p),
        ("seg_len", c_i32p),
        ("g_off", C.POINTER(C.c_int64)),
        ("et_off", C.POINTER(C.c_int64)),
        ("gvec_off", C.c_int64),
        ("n_vals", C.c_int64),
        ("e_off", C.POINTER(C.c_int64)),
        ("n_e", C.c_int64),
    ]


class ScreeningOptions(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int32),
        ("tolerance", C.c_double),
        ("step_size", C.c_double),
        ("step_drag", C.c_double),
    ]


# name -> (restype, argtypes); also the list of symbols the header declares
_CTX = C.c_void_p
SIGNATURES = {
    "tdgl_device_count": (C.c_int, []),
    "tdgl_version": (C.c_char_p, []),
    "tdgl_last_error": (C.c_char_p, [_CTX]),
    "tdgl_create": (C.c_int, [C.POINTER(_CTX), C.POINTER(MeshDesc), C.c_int]),
    "tdgl_destroy": (None, [_CTX]),
    "tdgl_synchronize": (C.c_int, [_CTX]),
    "tdgl_poisson_set_hierarchy": (C.c_int, [_CTX, C.POINTER(AmgLevel), C.c_int32, c_f64p]),
    "tdgl_set_poisson_options": (C.c_int, [_CTX, C.POINTER(PoissonOptions)]),
    "tdgl_get_poisson_stats": (C.c_int, [_CTX, C.POINTER(C.c_int64)]),
    "tdgl_get_pcg_prediction_stats": (C.c_int, [_CTX, C.POINTER(C.c_int64), c_f64p]),
    "tdgl_get_guess_stats": (C.c_int, [_CTX, C.POINTER(C.c_int32), c_f64p]),
    "tdgl_get_guess_gram": (C.c_int, [_CTX, C.POINTER(C.c_int32), c_f64p]),
    "tdgl_host_solve_gram": (C.c_int, [C.c_int32, c_f64p, c_f64p, C.c_double, c_f64p, C.POINTER(C.c_int32)]),
    "tdgl_poisson_set_fused_level": (
        C.c_int, [_CTX, C.c_int32, c_i32p, c_i32p, c_f64p, c_i32p, c_i32p, c_f64p, c_f64p]
    ),
    "tdgl_poisson_set_fused_restriction": (
        C.c_int, [_CTX, C.c_int64, C.c_int64, c_i32p, c_i32p, c_f64p, C.c_double]
    ),
    "tdgl_poisson_set_collapsed_level": (C.c_int, [_CTX, C.c_int32, c_i32p, c_i32p, c_f64p]),
    "tdgl_poisson_set_collapsed_up": (C.c_int, [_CTX, C.c_int32, c_i32p, c_i32p, c_f64p, c_i32p, c_i32p, c_f64p]),
    "tdgl_poisson_set_collapsed_tail": (C.c_int, [_CTX, C.POINTER(CollapsedTail)]),
    "tdgl_poisson_set_dense_inverse": (C.c_int, [_CTX, c_f64p, C.c_int64]),
    "tdgl_poisson_build_dense_inverse": (C.c_int, [_CTX, c_f64p]),
    "tdgl_poisson_set_substructure": (C.c_int, [_CTX, C.POINTER(Substructure), c_f64p]),
    "tdgl_poisson_set_substructure_inner": (C.c_int, [_CTX, C.POINTER(Substructure), c_f64p]),
    "tdgl_direct_switching": (C.c_int, [_CTX, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "tdgl_poisson_set_substructure_coupling": (C.c_int, [_CTX, C.c_int32, c_i32p, c_i32p, c_f64p]),
    "tdgl_poisson_set_substructure_precond": (C.c_int, [_CTX, c_i32p, C.c_int32, c_f64p, c_f64p]),
    "tdgl_poisson_precond_choice": (C.c_int, [_CTX, C.c_int32]),
    "tdgl_poisson_schur_begin": (C.c_int, [_CTX, C.POINTER(SchurPiece)]),
    "tdgl_poisson_schur_complement": (C.c_int, [_CTX, c_f64p]),
    "tdgl_poisson_schur_finish": (C.c_int, [_CTX, c_f64p, C.c_int32, c_f64p, c_f64p]),
    "tdgl_poisson_set_precond_times": (C.c_int, [_CTX, C.c_double, C.c_double]),
    "tdgl_get_precond_direct_stats": (C.c_int, [_CTX, C.POINTER(C.c_int64), c_f64p, C.c_int32]),
    "tdgl_get_precond_direct_layout": (C.c_int, [_CTX, C.POINTER(C.c_int32)]),
    "tdgl_poisson_set_substructure_layout": (C.c_int, [_CTX, C.c_int32]),
    "tdgl_poisson_build_substructure": (C.c_int, [_CTX, C.POINTER(SubstructurePlan), c_f64p]),
    "tdgl_set_halo_plan": (C.c_int, [_CTX, C.POINTER(HaloPlan)]),
    "tdgl_set_deep_halo_plan": (C.c_int, [_CTX, C.POINTER(DeepHaloPlan)]),
    "tdgl_comm_unique_id": (C.c_int, [C.c_char_p]),
    "tdgl_comm_init_rccl": (C.c_int, [_CTX, C.c_char_p]),
    "tdgl_comm_ipc_export": (C.c_int, [_CTX, C.c_char_p, c_i64p, C.c_int64]),
    "tdgl_comm_init_ipc": (C.c_int, [_CTX, C.c_char_p, c_i64p]),
    "tdgl_comm_ipc_set_timeout": (C.c_int, [_CTX, C.c_double]),
    "tdgl_comm_test_halo": (C.c_int, [_CTX, c_f64p, C.c_int32, C.c_int32]),
    "tdgl_comm_test_allreduce": (C.c_int, [_CTX, c_f64p, C.c_int64, C.c_int32, C.c_int32]),
    "tdgl_comm_init_callbacks": (C.c_int, [_CTX, HALO_FN, ALLREDUCE_FN, C.c_void_p]),
    "tdgl_set_comm_overlap": (C.c_int, [_CTX, C.c_int32]),
    "tdgl_get_comm_stats": (C.c_int, [_CTX, C.POINTER(C.c_int64), C.c_int32]),
    "tdgl_get_comm_overlap": (C.c_int, [_CTX, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "tdgl_set_link_exponents": (C.c_int, [_CTX, c_f64p]),
    "tdgl_set_link_exponents_base": (C.c_int, [_CTX, c_f64p, C.c_double]),
    "tdgl_update_link_scale": (C.c_int, [_CTX, C.c_double, C.c_double]),
    "tdgl_set_link_ramp": (C.c_int, [_CTX, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double]),
    "tdgl_get_link_scale": (C.c_int, [_CTX, c_f64p]),
    "tdgl_update_link_exponents": (C.c_int, [_CTX, c_f64p, C.c_double]),
    "tdgl_set_epsilon": (C.c_int, [_CTX, c_f64p]),
    "tdgl_set_mu_boundary": (C.c_int, [_CTX, c_f64p]),
    "tdgl_set_mu_boundary_table": (C.c_int, [_CTX, C.c_int32, c_f64p, C.c_int32, c_i32p, c_i32p, c_f64p]),
    "tdgl_set_epsilon_table": (C.c_int, [_CTX, c_f64p, C.c_int32, c_f64p, c_f64p]),
    "tdgl_set_state": (C.c_int, [_CTX, c_f64p, c_f64p]),
    "tdgl_set_controller": (C.c_int, [_CTX, C.POINTER(Controller)]),
    "tdgl_set_probes": (C.c_int, [_CTX, c_i32p, C.c_int32]),
    "tdgl_host_mean_tail": (C.c_double, [c_f64p, C.c_int64, C.c_int32]),
    "tdgl_begin_stage": (C.c_int, [_CTX]),
    "tdgl_run": (
        C.c_int,
        [_CTX, C.c_int64, C.c_double, c_f64p, c_f64p, c_f64p, c_i32p,
         C.POINTER(C.c_int64), C.POINTER(C.c_int32), c_i32p],
    ),
    "tdgl_set_screening": (
        C.c_int, [_CTX, C.POINTER(ScreeningOptions), c_f64p, c_f64p, c_f64p]
    ),
    "tdgl_set_screening_distributed": (
        C.c_int, [_CTX, C.POINTER(ScreeningOptions), c_f64p, c_f64p, C.POINTER(C.c_int64), c_f64p]
    ),
    "tdgl_set_induced_vector_potential": (C.c_int, [_CTX, c_f64p]),
    "tdgl_get_induced_vector_potential": (C.c_int, [_CTX, c_f64p]),
    "tdgl_induced_vector_potential": (C.c_int, [_CTX, c_f64p, c_f64p]),
    "tdgl_get_step_stats": (C.c_int, [_CTX, C.POINTER(C.c_int64), C.c_int32]),
    "tdgl_get_direct_stats": (C.c_int, [_CTX, c_f64p, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "tdgl_set_direct_guard": (C.c_int, [_CTX, C.c_double]),
    "tdgl_get_loop_state": (
        C.c_int,
        [_CTX, C.POINTER(C.c_int64), c_f64p, c_f64p, c_f64p],
    ),
    "tdgl_set_loop_state": (C.c_int, [_CTX, C.c_int64, C.c_double, C.c_double]),
    "tdgl_get_controller_state": (C.c_int, [_CTX, c_f64p, c_f64p, C.c_int64, C.POINTER(C.c_int64)]),
    "tdgl_set_controller_state": (C.c_int, [_CTX, C.c_double, c_f64p, C.c_int64]),
    "tdgl_get_state": (C.c_int, [_CTX, c_f64p, c_f64p, c_f64p, c_f64p]),
    "tdgl_apply_psi_laplacian": (C.c_int, [_CTX, c_f64p, c_f64p]),
    "tdgl_supercurrent": (C.c_int, [_CTX, c_f64p, c_f64p]),
    "tdgl_psi_update": (
        C.c_int, [_CTX, c_f64p, c_f64p, C.c_double, c_f64p, c_f64p, C.POINTER(C.c_int32)]
    ),
    "tdgl_poisson_rhs": (C.c_int, [_CTX, c_f64p, c_f64p]),
    "tdgl_poisson_solve": (C.c_int, [_CTX, c_f64p, c_f64p, C.POINTER(C.c_int32), c_f64p]),
    "tdgl_normal_current": (C.c_int, [_CTX, c_f64p, c_f64p]),
    "tdgl_apply_psi_gradient": (C.c_int, [_CTX, c_f64p, c_f64p]),
    "tdgl_apply_divergence": (C.c_int, [_CTX, c_f64p, c_f64p]),
    "tdgl_apply_mu_laplacian": (C.c_int, [_CTX, c_f64p, c_f64p]),
    "tdgl_apply_mu_boundary_laplacian": (C.c_int, [_CTX, c_f64p, c_f64p]),
    "tdgl_vcycle": (C.c_int, [_CTX, c_f64p, c_f64p]),
    "tdgl_guess_dots": (C.c_int, [_CTX, C.c_int32, C.c_int64, c_f64p, c_f64p, C.c_int32, c_f64p]),
    "tdgl_time_kernel": (C.c_int, [_CTX, C.c_int32, C.c_int32, c_f64p]),
    "tdgl_profile_enable": (C.c_int, [_CTX, C.c_int32]),
    "tdgl_profile_read_pcg": (C.c_int, [_CTX, C.POINTER(C.c_int64), c_f64p]),
    "tdgl_profile_read_direct": (C.c_int, [_CTX, C.POINTER(C.c_int64), c_f64p]),
    "tdgl_profile_read": (C.c_int, [_CTX, C.POINTER(C.c_int64), c_f64p]),
    "tdgl_profile_event_overhead": (C.c_int, [_CTX, C.c_int32, c_f64p]),
    "tdgl_get_precond_storage": (C.c_int, [_CTX, C.POINTER(C.c_int32)]),
}

_lib = None


class TDGLLibraryError(RuntimeError):
    pass


def load():
    """Load libtdgl_hip.so (once).  Raises TDGLLibraryError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TDGLLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g;"
            " g.build()'` (hipcc --offload-arch=gfx950).  tdgl_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def device_count() -> int:
    return int(load().tdgl_device_count())


def require_gpu():
    if device_count() < 1:
        raise TDGLLibraryError(
            "No HIP device is visible: tdgl_amd runs on MI355X (gfx950) only and has no CPU"
            " fallback."
        )


# ---- array marshalling helpers ---------------------------------------------------------
def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def c128(a):
    return np.ascontiguousarray(a, dtype=np.complex128)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def p_f64(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"] and a.dtype in (np.float64, np.complex128)
    return a.ctypes.data_as(c_f64p)


def p_i32(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"] and a.dtype == np.int32
    return a.ctypes.data_as(c_i32p)


def check(status, ctx=None):
    """Translate a tdgl_status into the exception the reference would raise."""
    if status == TDGL_OK:
        return
    msg = load().tdgl_last_error(ctx)
    msg = msg.decode() if msg else f"tdgl status {status}"
    if status == TDGL_ERR_ARG:
        raise ValueError(msg)
    raise RuntimeError(msg)  # PSI_RETRIES carries the reference's wording (solver.py:479-483)

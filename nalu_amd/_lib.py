"""ctypes binding of libnalu_gpu.so (include/nalu_gpu.h).  Fails loudly if the HIP
library is missing: there is no CPU fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NALU_GPU_LIB", os.path.join(_HERE, "libnalu_gpu.so"))

SCATTER = {"rows": 0, "colored": 1, "atomic": 2}
METHOD = {"cg": 0, "gmres": 1, "biCgStab": 2, "bicgstab": 2, "tfqmr": 3}
PRECOND = {"none": 0, "jacobi": 1, "default": 1, "sgs": 2, "chebyshev": 3, "mt_sgs": 4}
PECLET = {"classic": 0, "tanh": 1}
# DofStatus (include/TpetraLinearSystem.h:49-55)
DS_NotSet, DS_SkippedDOF, DS_OwnedDOF, DS_SharedNotOwnedDOF, DS_GhostedDOF = 0, 2, 4, 8, 16
TIMERS = {"zero": 0, "assemble_edges": 1, "spmv": 2, "solve": 3, "precond": 4, "bc": 5}

# every symbol include/nalu_gpu.h declares (tests check the library exports them all)
SYMBOLS = [
    "nalu_gpu_last_error", "nalu_gpu_version", "nalu_gpu_create", "nalu_gpu_destroy",
    "nalu_gpu_begin_construction", "nalu_gpu_get_dof_status", "nalu_gpu_build_node_graph",
    "nalu_gpu_build_edge_to_node_graph", "nalu_gpu_build_elem_to_node_graph",
    "nalu_gpu_build_dirichlet_node_graph", "nalu_gpu_finalize", "nalu_gpu_get_sizes",
    "nalu_gpu_get_graph", "nalu_gpu_get_matrix_values", "nalu_gpu_get_rhs",
    "nalu_gpu_get_solution", "nalu_gpu_set_matrix_values", "nalu_gpu_set_rhs",
    "nalu_gpu_zero_system", "nalu_gpu_sum_into", "nalu_gpu_scatter_plan_create",
    "nalu_gpu_scatter_plan_destroy", "nalu_gpu_scatter_plan_colors", "nalu_gpu_sum_into_batch",
    "nalu_gpu_assemble_continuity_edges",
    "nalu_gpu_assemble_scalar_edges", "nalu_gpu_assemble_scalar_mass_be",
    "nalu_gpu_apply_dirichlet", "nalu_gpu_reset_rows", "nalu_gpu_load_complete",
    "nalu_gpu_solve", "nalu_gpu_spmv", "nalu_gpu_dot", "nalu_gpu_axpy", "nalu_gpu_norm2",
    "nalu_gpu_compute_mdot_edges", "nalu_gpu_nodal_grad_edges",
    "nalu_gpu_field_axpby", "nalu_gpu_project_nodal_velocity", "nalu_gpu_compute_geometry_hex8", "nalu_gpu_set_halo_plan", "nalu_gpu_set_halo_plan_indexed",
    "nalu_gpu_set_comm", "nalu_gpu_set_callback_error", "nalu_gpu_get_halo_plan", "nalu_gpu_set_assembly_mode", "nalu_gpu_set_edge_ownership", "nalu_gpu_get_comm_buffers", "nalu_gpu_halo_exchange",
    "nalu_gpu_set_profiling", "nalu_gpu_get_timer", "nalu_gpu_get_timer_launches", "nalu_gpu_reset_timers",
    "nalu_gpu_synchronize", "nalu_gpu_get_layout_info", "nalu_gpu_suggest_edge_order", "nalu_gpu_check_system", "nalu_gpu_set_debug", "nalu_gpu_dot_static", "nalu_gpu_get_sgs_schedule",
    "nalu_gpu_update_and_clip", "nalu_gpu_share_plan_create", "nalu_gpu_share_plan_destroy", "nalu_gpu_parallel_sum",
]


class Vec3(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("z", C.c_void_p)]


class ContinuityFields(C.Structure):
    _fields_ = [("coordinates", Vec3), ("dpdx", Vec3), ("velocity", Vec3),
                ("pressure", C.c_void_p), ("density", C.c_void_p), ("edge_area", Vec3)]


class ScalarFields(C.Structure):
    _fields_ = [("coordinates", Vec3), ("dqdx", Vec3), ("velocity", Vec3),
                ("scalar_q", C.c_void_p), ("density", C.c_void_p),
                ("diff_flux_coeff", C.c_void_p), ("edge_area", Vec3),
                ("mass_flow_rate", C.c_void_p)]


class ScalarOptions(C.Structure):
    _fields_ = [("alpha", C.c_double), ("alpha_upw", C.c_double), ("ho_upwind", C.c_double),
                ("use_limiter", C.c_int), ("peclet_kind", C.c_int),
                ("hybrid_factor", C.c_double), ("tanh_c1", C.c_double), ("tanh_c2", C.c_double)]


class SolverConfig(C.Structure):
    _fields_ = [("method", C.c_int), ("precond", C.c_int), ("tolerance", C.c_double),
                ("max_iterations", C.c_int), ("kspace", C.c_int), ("check_every", C.c_int)]


class SolveInfo(C.Structure):
    _fields_ = [("iters", C.c_int), ("final_resid", C.c_double), ("rhs_norm2", C.c_double),
                ("precond_seconds", C.c_double), ("solve_seconds", C.c_double),
                ("history_len", C.c_int)]


class CheckInfo(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("lhs_nan_rows", "rhs_nan_rows", "zero_rows", "first_lhs_nan_row",
                                          "first_rhs_nan_row", "first_zero_row")]


class LayoutInfo(C.Structure):
    _fields_ = [("n_slices", C.c_int64), ("sell_slots", C.c_int64), ("max_slice_width", C.c_int),
                ("rows_computed_columns", C.c_int64), ("rows_computed_tags", C.c_int64),
                ("assembly_variant", C.c_int), ("xcd_run", C.c_int),
                ("march_runs", C.c_int64), ("march_slices", C.c_int64), ("march_longest", C.c_int)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)
EXCHANGE_ROWS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p,
                               C.POINTER(C.c_int64), C.c_void_p)


class Comm(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("allreduce_sum", ALLREDUCE_FN),
                ("exchange_begin", EXCHANGE_FN), ("exchange_end", EXCHANGE_FN),
                ("exchange_rows", EXCHANGE_ROWS_FN)]


_lib = None


class NaluGpuError(RuntimeError):
    pass


def load():
    """Load libnalu_gpu.so; raises if it was not built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NaluGpuError(
            f"{LIB_PATH} is missing: build it with `make -C nalu_amd/csrc` "
            "(or __graft_entry__.build()); nalu_amd has no CPU fallback")
    # PyTorch-ROCm ships its own libamdhip64.so.7; import it first so that this library binds
    # to the same HIP runtime instance (two runtimes in one process hide the GPU from torch).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    lib.nalu_gpu_last_error.restype = C.c_char_p
    vp, i64, dbl, ci = C.c_void_p, C.c_int64, C.c_double, C.c_int
    P = C.POINTER
    sig = {
        "nalu_gpu_create": [P(vp), ci, ci, vp],
        "nalu_gpu_destroy": [vp],
        "nalu_gpu_begin_construction": [vp, ci, ci, ci, vp, vp, ci],
        "nalu_gpu_get_dof_status": [vp, i64, vp, vp],
        "nalu_gpu_build_node_graph": [vp, i64, vp],
        "nalu_gpu_build_edge_to_node_graph": [vp, i64, vp],
        "nalu_gpu_build_elem_to_node_graph": [vp, i64, ci, vp],
        "nalu_gpu_build_dirichlet_node_graph": [vp, i64, vp],
        "nalu_gpu_finalize": [vp],
        "nalu_gpu_get_sizes": [vp, P(i64), P(i64), P(i64)],
        "nalu_gpu_get_graph": [vp, vp, vp],
        "nalu_gpu_get_matrix_values": [vp, vp],
        "nalu_gpu_get_rhs": [vp, vp],
        "nalu_gpu_get_solution": [vp, vp],
        "nalu_gpu_set_matrix_values": [vp, vp],
        "nalu_gpu_set_rhs": [vp, vp],
        "nalu_gpu_zero_system": [vp],
        "nalu_gpu_sum_into": [vp, ci, vp, vp, vp],
        "nalu_gpu_scatter_plan_create": [vp, i64, ci, vp, P(vp)],
        "nalu_gpu_scatter_plan_destroy": [vp],
        "nalu_gpu_scatter_plan_colors": [vp, P(ci)],
        "nalu_gpu_sum_into_batch": [vp, vp, vp, vp, ci],
        "nalu_gpu_assemble_continuity_edges": [vp, P(ContinuityFields), dbl, dbl, dbl, dbl, ci],
        "nalu_gpu_assemble_scalar_edges": [vp, P(ScalarFields), P(ScalarOptions), ci],
        "nalu_gpu_assemble_scalar_mass_be": [vp, vp, vp, vp, vp, vp, dbl],
        "nalu_gpu_apply_dirichlet": [vp, i64, vp, vp, vp],
        "nalu_gpu_reset_rows": [vp, i64, vp],
        "nalu_gpu_load_complete": [vp],
        "nalu_gpu_solve": [vp, P(SolverConfig), vp, P(SolveInfo), vp, ci],
        "nalu_gpu_spmv": [vp, vp, vp],
        "nalu_gpu_dot": [vp, vp, vp, i64, P(dbl)],
        "nalu_gpu_axpy": [vp, dbl, vp, vp, i64],
        "nalu_gpu_norm2": [vp, vp, i64, P(dbl)],
        "nalu_gpu_compute_mdot_edges": [vp, P(ContinuityFields), dbl, dbl, dbl, dbl, vp],
        "nalu_gpu_nodal_grad_edges": [vp, Vec3, vp, vp, vp, vp, vp],
        "nalu_gpu_compute_geometry_hex8": [vp, i64, vp, Vec3, vp, vp, vp, vp],
        "nalu_gpu_field_axpby": [vp, i64, dbl, vp, dbl, vp],
        "nalu_gpu_project_nodal_velocity": [vp, i64, dbl, dbl, Vec3, Vec3, vp, vp, Vec3],
        "nalu_gpu_set_halo_plan": [vp, ci, vp, vp, vp, vp],
        "nalu_gpu_set_halo_plan_indexed": [vp, ci, vp, vp, vp, vp, vp],
        "nalu_gpu_set_comm": [vp, P(Comm)],
        "nalu_gpu_set_callback_error": [C.c_char_p],
        "nalu_gpu_get_halo_plan": [vp, P(ci), P(vp), P(vp), P(vp)],
        "nalu_gpu_set_assembly_mode": [vp, ci],
        "nalu_gpu_set_edge_ownership": [vp, i64, vp],
        "nalu_gpu_get_comm_buffers": [vp, P(vp), P(i64), P(vp), P(i64), P(vp), P(ci)],
        "nalu_gpu_halo_exchange": [vp, vp],
        "nalu_gpu_set_profiling": [vp, ci],
        "nalu_gpu_get_timer": [vp, ci, P(dbl), P(i64)],
        "nalu_gpu_get_timer_launches": [vp, ci, P(i64)],
        "nalu_gpu_reset_timers": [vp],
        "nalu_gpu_get_sgs_schedule": [vp, P(ci)],
        "nalu_gpu_synchronize": [vp],
        "nalu_gpu_get_layout_info": [vp, P(LayoutInfo)],
        "nalu_gpu_suggest_edge_order": [i64, vp, vp],
        "nalu_gpu_check_system": [vp, P(CheckInfo)],
        "nalu_gpu_set_debug": [vp, ci],
        "nalu_gpu_dot_static": [ci, vp, vp, vp, i64, vp],
        "nalu_gpu_update_and_clip": [vp, i64, vp, vp, vp, dbl, vp],
        "nalu_gpu_share_plan_create": [vp, vp, vp, vp, vp, i64, P(vp)],
        "nalu_gpu_share_plan_destroy": [vp],
        "nalu_gpu_parallel_sum": [vp, vp, ci, vp],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = ci
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise NaluGpuError(f"nalu_gpu error {rc}: {load().nalu_gpu_last_error().decode()}")

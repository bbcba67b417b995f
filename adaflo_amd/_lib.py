"""ctypes binding of include/adaflo_hip.h (the C-ABI drop-in boundary)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (ADAFLO_LIB_PATH: development only -- A/B runs of differently compiled engines, scripts/dev; never a CPU fallback)
LIB_PATH = os.environ.get("ADAFLO_LIB_PATH") or os.path.join(_HERE, "lib", "libadaflo_hip.so")

# status codes of include/adaflo_hip.h
ADAFLO_OK, ADAFLO_EINVAL, ADAFLO_ENOTINIT, ADAFLO_EHIP, ADAFLO_ENOMEM, ADAFLO_EUNSUPPORTED = 0, -1, -2, -3, -4, -5


class BrickDesc(C.Structure):
    _fields_ = [("dim", C.c_int), ("ncell", C.c_int * 3), ("h", C.c_double * 3),
                ("origin", C.c_double * 3), ("velocity_degree", C.c_int), ("ls_degree", C.c_int),
                ("velocity_constrained", C.c_uint32), ("pressure_constrained", C.c_uint32),
                ("ls_constrained", C.c_uint32), ("pressure_average_fix", C.c_int),
                ("device", C.c_int), ("stream", C.c_void_p)]


class IndexedDesc(C.Structure):
    _fields_ = [("device", C.c_int), ("stream", C.c_void_p), ("velocity_degree", C.c_int), ("pressure_average_fix", C.c_int),
                ("n_cells", C.c_int64), ("n_nodes_u", C.c_int64), ("n_nodes_p", C.c_int64),
                ("cell_nodes_u", C.POINTER(C.c_int)), ("cell_nodes_p", C.POINTER(C.c_int)),
                ("constrained_u", C.POINTER(C.c_ubyte)), ("constrained_p", C.POINTER(C.c_ubyte)),
                ("cell_extents", C.POINTER(C.c_double)), ("h", C.c_double * 3),
                ("n_colours", C.c_int), ("colour_offsets", C.POINTER(C.c_int64)),
                ("n_hanging_u", C.c_int64), ("n_hanging_p", C.c_int64),
                ("hanging_ptr_u", C.POINTER(C.c_int64)), ("hanging_ptr_p", C.POINTER(C.c_int64)),
                ("hanging_master_u", C.POINTER(C.c_int)), ("hanging_master_p", C.POINTER(C.c_int)),
                ("hanging_weight_u", C.POINTER(C.c_double)), ("hanging_weight_p", C.POINTER(C.c_double))]


class LSParams(C.Structure):
    _fields_ = [("epsilon_used", C.c_double), ("minimal_edge_length", C.c_double),
                ("time_step", C.c_double), ("weight", C.c_double), ("weight_old", C.c_double),
                ("weight_old_old", C.c_double), ("epsilon", C.c_double)]


class ForceParams(C.Structure):
    _fields_ = [("surface_tension", C.c_double), ("gravity", C.c_double), ("density", C.c_double),
                ("density_diff", C.c_double), ("viscosity", C.c_double), ("viscosity_diff", C.c_double),
                ("interpolate_grad_onto_pressure", C.c_int)]


class SolverControl(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("abs_tol", C.c_double), ("rel_tol", C.c_double)]


class SolverResult(C.Structure):
    _fields_ = [("iterations", C.c_int), ("converged", C.c_int), ("initial_residual", C.c_double),
                ("final_residual", C.c_double)]


class NSParams(C.Structure):
    _fields_ = [("physical_type", C.c_int), ("linearization", C.c_int), ("beta", C.c_double),
                ("tau_grad_div", C.c_double), ("density", C.c_double), ("viscosity", C.c_double),
                ("damping", C.c_double), ("density_diff", C.c_double), ("weight", C.c_double),
                ("weight_old", C.c_double), ("weight_old_old", C.c_double), ("tau1", C.c_double),
                ("extrap_old", C.c_double), ("extrap_old_old", C.c_double)]


_D = C.c_void_p  # device pointer
_CTX = C.c_void_p
_COMM = C.c_void_p


class CommUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


_I64P = C.POINTER(C.c_int64)
_INTP = C.POINTER(C.c_int)
class MinMaxAvg(C.Structure):
    """adaflo_min_max_avg = dealii::Utilities::MPI::MinMaxAvg"""
    _fields_ = [("sum", C.c_double), ("min", C.c_double), ("max", C.c_double), ("avg", C.c_double),
                ("min_index", C.c_int), ("max_index", C.c_int)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, _D, _I64P, _I64P, _INTP, C.c_int, _D, _I64P, _I64P, _INTP, C.c_int,
                          C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, _D, C.c_int, C.c_void_p)

# name -> (restype, argtypes); mirrors include/adaflo_hip.h one to one
SIGNATURES = {
    "adaflo_ctx_create": (C.c_int, [C.POINTER(BrickDesc), C.POINTER(_CTX)]),
    "adaflo_ctx_create_indexed": (C.c_int, [C.POINTER(IndexedDesc), C.POINTER(_CTX)]),
    "adaflo_ctx_destroy": (C.c_int, [_CTX]),
    "adaflo_last_error": (C.c_char_p, [_CTX]),
    "adaflo_synchronize": (C.c_int, [_CTX]),
    "adaflo_stream": (C.c_void_p, [_CTX]),
    "adaflo_set_stream": (C.c_int, [_CTX, C.c_void_p]),
    "adaflo_n_cells": (C.c_int64, [_CTX]),
    "adaflo_n_dofs_u": (C.c_int64, [_CTX]),
    "adaflo_n_dofs_p": (C.c_int64, [_CTX]),
    "adaflo_n_dofs_ls": (C.c_int64, [_CTX]),
    "adaflo_n_q_points_u": (C.c_int, [_CTX]),
    "adaflo_n_q_points_ls": (C.c_int, [_CTX]),
    "adaflo_malloc": (C.c_int, [_CTX, C.c_size_t, C.POINTER(C.c_void_p)]),
    "adaflo_free": (C.c_int, [_CTX, _D]),
    "adaflo_copy_h2d": (C.c_int, [_CTX, _D, C.c_void_p, C.c_size_t]),
    "adaflo_copy_d2h": (C.c_int, [_CTX, C.c_void_p, _D, C.c_size_t]),
    "adaflo_ns_set_params": (C.c_int, [_CTX, C.POINTER(NSParams)]),
    "adaflo_ns_set_linearization": (C.c_int, [_CTX, C.c_void_p, C.c_int]),
    "adaflo_ns_get_linearization": (C.c_int, [_CTX, C.c_void_p, C.c_int]),
    "adaflo_ns_set_coefficients": (C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "adaflo_ns_get_coefficients": (C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "adaflo_ns_fix_linearization_point": (C.c_int, [_CTX]),
    "adaflo_ns_vmult": (C.c_int, [_CTX, _D, _D, _D, _D]),
    "adaflo_ns_preconditioner_set_cheap_velocity_iterations": (C.c_int, [_CTX, C.c_int]),
    "adaflo_ns_vmult_phase": (C.c_int, [_CTX, _D, _D, _D, _D, C.c_int, C.c_uint]),
    "adaflo_ns_supports_phases": (C.c_int, [_CTX]),
    "adaflo_ns_residual": (C.c_int, [_CTX, _D, _D, _D, _D, _D, _D, _D, _D]),
    "adaflo_ns_velocity_vmult": (C.c_int, [_CTX, _D, _D]),
    "adaflo_ns_velocity_block_diagonal": (C.c_int, [_CTX, _D]),
    "adaflo_ns_divergence_vmult_add": (C.c_int, [_CTX, _D, _D, C.c_int]),
    "adaflo_ns_pressure_poisson_vmult": (C.c_int, [_CTX, _D, _D]),
    "adaflo_ns_pressure_mass_vmult": (C.c_int, [_CTX, _D, _D]),
    "adaflo_ns_pressure_convdiff_vmult": (C.c_int, [_CTX, _D, _D]),
    "adaflo_ns_apply_pressure_average_projection": (C.c_int, [_CTX, _D]),
    "adaflo_ns_pressure_mass_weight_add": (C.c_int, [_CTX, _D]),
    "adaflo_ns_apply_constrained_rows": (C.c_int, [_CTX, _D, _D, _D, _D]),
    "adaflo_ns_get_matvec_statistics": (C.c_int, [_CTX, C.POINTER(C.c_uint), C.POINTER(C.c_double)]),
    "adaflo_halo_transfer": (C.c_int, [_CTX, _D, _D, C.POINTER(C.c_int), C.c_int, C.c_int,
                                      C.POINTER(C.c_int), C.c_int]),
    "adaflo_halo_transfer_ordered": (C.c_int, [_CTX, _D, _D, C.POINTER(C.c_int), C.c_int, C.c_int,
                                              C.POINTER(C.c_int), C.c_int, C.c_int]),
    "adaflo_comm_get_unique_id": (C.c_int, [C.POINTER(CommUniqueId)]),
    "adaflo_comm_create": (C.c_int, [_CTX, C.POINTER(CommUniqueId), C.c_int, C.c_int, _INTP, C.c_int, C.POINTER(_COMM)]),
    "adaflo_comm_create_custom": (C.c_int, [_CTX, C.c_int, C.c_int, _INTP, EXCHANGE_FN, ALLREDUCE_FN, C.c_void_p, C.c_int,
                                            C.POINTER(_COMM)]),
    "adaflo_comm_destroy": (C.c_int, [_COMM]),
    "adaflo_comm_last_error": (C.c_char_p, [_COMM]),
    "adaflo_comm_interface_faces": (C.c_uint, [_COMM]),
    "adaflo_comm_update_ghost_values": (C.c_int, [_COMM, _D, _D]),
    "adaflo_comm_compress_add": (C.c_int, [_COMM, _D, _D]),
    "adaflo_ns_vmult_distributed": (C.c_int, [_CTX, _COMM, _D, _D, _D, _D, C.c_int]),
    "adaflo_comm_force_phased_schedule": (C.c_int, [_COMM, C.c_int]),
    "adaflo_comm_matvec_statistics": (C.c_int, [_COMM, C.POINTER(C.c_uint), C.POINTER(MinMaxAvg)]),
    "adaflo_comm_set_phase_timing": (C.c_int, [_COMM, C.c_int]),
    "adaflo_comm_phase_statistics": (C.c_int, [_COMM, C.POINTER(C.c_uint), C.POINTER(C.c_double)]),
    "adaflo_ls_set_params": (C.c_int, [_CTX, C.POINTER(LSParams)]),
    "adaflo_ls_set_diagonal": (C.c_int, [_CTX, _D]),
    "adaflo_ls_set_evaluated_convection": (C.c_int, [_CTX, C.c_void_p, C.c_int]),
    "adaflo_ls_get_evaluated_convection": (C.c_int, [_CTX, C.c_void_p, C.c_int]),
    "adaflo_ls_set_evaluated_normal": (C.c_int, [_CTX, C.c_void_p, C.c_int]),
    "adaflo_ls_get_evaluated_normal": (C.c_int, [_CTX, C.c_void_p, C.c_int]),
    "adaflo_ls_advance_concentration_vmult": (C.c_int, [_CTX, _D, _D]),
    "adaflo_ls_advance_concentration_rhs": (C.c_int, [_CTX, _D, _D, _D, _D, _D, C.c_int]),
    "adaflo_ls_set_convection_stabilization": (C.c_int, [_CTX, C.c_int, C.c_double, C.c_uint]),
    "adaflo_ls_set_artificial_viscosities": (C.c_int, [_CTX, C.c_void_p, C.c_int]),
    "adaflo_ls_get_artificial_viscosities": (C.c_int, [_CTX, C.c_void_p, C.c_int]),
    "adaflo_ls_max_velocity": (C.c_int, [_CTX, _D, C.POINTER(C.c_double)]),
    "adaflo_ls_advance_concentration_rhs_stabilized": (C.c_int, [_CTX, _D, _D, _D, _D, _D, _D, _D, C.c_int, C.c_double,
                                                                C.c_double]),
    "adaflo_ls_stabilization_boundary_term": (C.c_int, [_CTX, _D, _D, C.c_double]),
    "adaflo_ls_reinitialization_vmult": (C.c_int, [_CTX, _D, _D, C.c_int]),
    "adaflo_ls_reinitialization_rhs": (C.c_int, [_CTX, _D, _D, _D, C.c_int, C.c_int]),
    "adaflo_ls_compute_normal_vmult": (C.c_int, [_CTX, _D, _D]),
    "adaflo_ls_compute_normal_rhs": (C.c_int, [_CTX, _D, _D]),
    "adaflo_ls_compute_curvature_vmult": (C.c_int, [_CTX, _D, _D, C.c_int]),
    "adaflo_ls_compute_curvature_rhs": (C.c_int, [_CTX, _D, _D]),
    "adaflo_set_kernel_variant": (C.c_int, [_CTX, C.c_int]),
    "adaflo_has_kernel_variant": (C.c_int, [C.c_int]),
    "adaflo_ls_compute_heaviside": (C.c_int, [_CTX, _D, _D, C.c_double]),
    "adaflo_ls_curvature_correction": (C.c_int, [_CTX, _D, _D]),
    "adaflo_ls_projection_vmult": (C.c_int, [_CTX, _D, _D]),
    "adaflo_ls_compute_force": (C.c_int, [_CTX, _D, _D, _D, C.POINTER(ForceParams)]),
    "adaflo_vector_gather": (C.c_int, [_CTX, _D, _D, C.c_void_p, C.c_int64]),
    "adaflo_vector_scatter": (C.c_int, [_CTX, _D, _D, C.c_void_p, C.c_int64, C.c_int]),
    "adaflo_vector_fill": (C.c_int, [_CTX, _D, C.c_double, C.c_int64]),
    "adaflo_vector_sadd": (C.c_int, [_CTX, _D, C.c_double, C.c_double, _D, C.c_int64]),
    "adaflo_vector_dot": (C.c_int, [_CTX, _D, _D, C.c_int64, C.POINTER(C.c_double)]),
    "adaflo_ls_mass_matrix_diagonal": (C.c_int, [_CTX, _D]),
    "adaflo_invert_diagonal": (C.c_int, [_CTX, _D, _D, C.c_int64]),
    "adaflo_fdm_apply": (C.c_int, [_CTX, C.c_int, _D, _D, C.c_double, C.c_double]),
    "adaflo_fdm_apply_sum": (C.c_int, [_CTX, C.c_int, _D, _D, C.c_double, C.c_double, C.c_double, C.c_double]),
    "adaflo_ls_projection_solve": (C.c_int, [_CTX, _D, _D, C.c_int]),
    "adaflo_ns_set_iterations_before_inner_solvers": (C.c_int, [_CTX, C.c_int]),
    "adaflo_ns_preconditioner_set_inner": (C.c_int, [_CTX, C.c_int]),
    "adaflo_ns_preconditioner_statistics": (C.c_int, [_CTX, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "adaflo_ns_preconditioner_setup": (C.c_int, [_CTX]),
    "adaflo_ns_preconditioner_vmult": (C.c_int, [_CTX, _D, _D, _D, _D]),
    "adaflo_ns_solve_system": (C.c_int, [_CTX, _D, _D, _D, _D, C.POINTER(SolverControl), C.c_int,
                                         C.POINTER(SolverResult)]),
    "adaflo_solve": (C.c_int, [_CTX, C.c_int, C.c_int, _D, _D, _D, C.POINTER(SolverControl),
                               C.POINTER(SolverResult)]),
    "adaflo_get_kernel_statistics": (C.c_int, [_CTX, C.POINTER(C.c_uint), C.POINTER(C.c_double)]),
    "adaflo_set_timing": (C.c_int, [_CTX, C.c_int]),
    "adaflo_set_q2_chunk": (C.c_int, [_CTX, C.c_int]),
    "adaflo_set_q2_lazy_state": (C.c_int, [_CTX, C.c_int]),
    "adaflo_set_hox_chunk": (C.c_int, [_CTX, C.c_int]),
    "adaflo_set_q2_state_pad": (C.c_int, [_CTX, C.c_int]),
}

_lib = None


def load():
    """Load the HIP engine; raises (no fallback) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "HIP engine %s is missing: run `python -m adaflo_amd.build` (hipcc, gfx950). "
                "There is no CPU fallback." % LIB_PATH)
        # One HIP runtime per process: the PyTorch-ROCm wheel ships its own libamdhip64 and the
        # engine links the one under /opt/rocm.  If the engine's copy initialises the GPU first,
        # torch later loads a second runtime that sees no device ("No HIP GPUs are available").
        # Importing torch first makes the dynamic linker resolve the engine's libamdhip64.so.7 to
        # the copy that is already loaded.  (ADAFLO_NO_TORCH=1: engine-only processes.)
        if os.environ.get("ADAFLO_NO_TORCH") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class AdafloError(RuntimeError):
    pass


class CtxHandle(C.c_void_p):
    """the engine context as the operator object owns it: vectors and level-set operators keep a
    reference to this handle and stop calling into the library once the context is destroyed"""
    alive = False


def check(ctx, code):
    if code != 0:
        msg = load().adaflo_last_error(ctx)
        raise AdafloError("adaflo_hip error %d: %s" % (code, msg.decode() if msg else "?"))

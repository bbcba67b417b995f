"""Host-side mirror of the Krylov callers of the operators (device-resident vectors).

Names follow deal.II / adaflo so that code reads like the reference:

    control = ReductionControl(2000, 1e-50, 1e-6)         # level_set_okz_reinitialization.cc:333
    cg = SolverCG(control)
    cg.solve(ReinitializationMatrix(ops, diffuse), increment, rhs, preconditioner)

The iterations run inside libadaflo_hip.so (csrc/krylov.hip) on the engine's stream."""
import ctypes as C

from . import _lib

OPERATORS = {"advance_concentration": 0, "reinitialization": 1, "reinitialization_diffuse": 2,
             "normal": 3, "curvature": 4, "pressure_mass": 5, "pressure_poisson": 6, "velocity": 7,
             "projection": 8}


class NoConvergence(RuntimeError):
    """SolverControl::NoConvergence"""

    def __init__(self, last_step, last_residual):
        super().__init__("Iterative method reported convergence failure in step %d, residual %g"
                         % (last_step, last_residual))
        self.last_step, self.last_residual = last_step, last_residual


class ReductionControl:
    """ReductionControl(n, tol, reduce)"""

    def __init__(self, max_steps=100, tolerance=1e-10, reduce=1e-2):
        self.max_steps, self.tolerance, self.reduce = max_steps, tolerance, reduce
        self._last_step, self._initial, self._last = 0, 0.0, 0.0

    def last_step(self):
        return self._last_step

    def initial_value(self):
        return self._initial

    def last_value(self):
        return self._last


class DiagonalPreconditioner:
    """adaflo::DiagonalPreconditioner (source/diagonal_preconditioner.cc): pointwise inverse of a
    diagonal vector, entries below 1e-10 of the largest one replaced by 1"""

    def __init__(self, diagonal_vector):
        from .vectors import DeviceVector
        self.diagonal_vector = diagonal_vector
        ctx = diagonal_vector.ctx
        self.inverse_diagonal_vector = DeviceVector(ctx, diagonal_vector.n)
        _lib.check(ctx, _lib.load().adaflo_invert_diagonal(ctx, self.inverse_diagonal_vector.ptr,
                                                          diagonal_vector.ptr, diagonal_vector.n))

    def get_vector(self):
        return self.diagonal_vector


class _Matrix:
    """operator handle: (engine context, operator id)"""

    def __init__(self, ctx, op):
        self._ctx, self.op = ctx, OPERATORS[op]


class AdvanceConcentrationMatrix(_Matrix):
    def __init__(self, ops):
        super().__init__(ops._ctx, "advance_concentration")


class ReinitializationMatrix(_Matrix):
    def __init__(self, ops, diffuse_only):
        super().__init__(ops._ctx, "reinitialization_diffuse" if diffuse_only else "reinitialization")


class ComputeNormalMatrix(_Matrix):
    def __init__(self, ops):
        super().__init__(ops._ctx, "normal")


class ComputeCurvatureMatrix(_Matrix):
    def __init__(self, ops):
        super().__init__(ops._ctx, "curvature")


class ProjectionMatrix(_Matrix):
    """the scalar projection matrix of LevelSetOKZSolver (level_set_okz.cc:262-312) the production
    curvature solve uses (compute_curvature.cc:355), applied matrix-free"""

    def __init__(self, ops):
        super().__init__(ops._ctx, "projection")


class PressureMassMatrix(_Matrix):
    def __init__(self, ns_matrix):
        super().__init__(ns_matrix._require(), "pressure_mass")


class PressurePoissonMatrix(_Matrix):
    def __init__(self, ns_matrix):
        super().__init__(ns_matrix._require(), "pressure_poisson")


class VelocityMatrix(_Matrix):
    def __init__(self, ns_matrix):
        super().__init__(ns_matrix._require(), "velocity")


class _Solver:
    method = None

    def __init__(self, control):
        self.control = control

    def solve(self, matrix, x, b, preconditioner=None):
        c = self.control
        ctl = _lib.SolverControl(c.max_steps, c.tolerance, c.reduce)
        res = _lib.SolverResult()
        inv = preconditioner.inverse_diagonal_vector.ptr if preconditioner is not None else None
        _lib.check(matrix._ctx, _lib.load().adaflo_solve(matrix._ctx, matrix.op, self.method, x.ptr, b.ptr,
                                                        inv, C.byref(ctl), C.byref(res)))
        c._last_step, c._initial, c._last = res.iterations, res.initial_residual, res.final_residual
        if not res.converged:
            raise NoConvergence(res.iterations, res.final_residual)


class SolverCG(_Solver):
    method = 0


class SolverBicgstab(_Solver):
    method = 1

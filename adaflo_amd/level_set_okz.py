"""Host-side mirrors of adaflo's four level-set operator classes
(include/adaflo/level_set_okz_{advance_concentration,reinitialization,compute_normal,
compute_curvature}.h): same method names and argument meaning, forwarding to the C ABI.

In the reference the operators hold references to the vectors they work on
(constructor arguments, e.g. level_set_okz_reinitialization.h:74-104); here the vectors
are passed per call.  All four share one engine context (one MatrixFree in the reference,
source/two_phase_base.cc:239-275) created by `LevelSetOperators`."""
import ctypes as C

import numpy as np

from . import _lib
from .vectors import DeviceVector


def _projection_solve(solver, dst, rhs, n_blocks):
    """exact inverse of the projection matrix (adaflo_ls_projection_solve); False when the engine refuses it -- a
    level-set space with constrained faces, ADAFLO_EUNSUPPORTED -- so that the caller keeps the CG path of the
    reference; any other error is raised"""
    code = solver._lib.adaflo_ls_projection_solve(solver._ctx, dst.ptr, rhs.ptr, n_blocks)
    if code == _lib.ADAFLO_EUNSUPPORTED:
        return False
    _lib.check(solver._ctx, code)
    return True


class LevelSetOperators:
    """owns the engine context for the level-set spaces on a brick (FE_Q_iso_Q1(ls_degree))"""

    # normal / curvature projections: False = CG with the mass-diagonal preconditioner to the reference's
    # tolerances (1e-7 / 1e-8), True = exact inverse of the projection matrix by fast diagonalisation
    # (adaflo_ls_projection_solve; unconstrained level-set space only)
    exact_projection = False

    def __init__(self, mesh, ls_degree, velocity_degree=2, constrained_faces=(), device=0, stream=None,
                 dirichlet_faces_u=(), navier_stokes_matrix=None):
        """navier_stokes_matrix: share the engine context of an initialised NavierStokesMatrix that
        was constructed with ls_degree = this degree, so that compute_force writes the density /
        viscosity arrays of THAT operator (LevelSetOKZSolver holds a reference to navier_stokes)"""
        self._lib = _lib.load()
        self.mesh, self.s, self.k = mesh, ls_degree, velocity_degree
        self._owns_ctx = navier_stokes_matrix is None
        self._owner = navier_stokes_matrix     # keeps the shared context's owner alive
        if navier_stokes_matrix is not None:
            self._ctx = navier_stokes_matrix._require()
            assert self._lib.adaflo_n_dofs_ls(self._ctx) > 0, "NavierStokesMatrix was built without ls_degree"
            self.n_dofs = self._lib.adaflo_n_dofs_ls(self._ctx)
            self.n_q = self._lib.adaflo_n_q_points_ls(self._ctx)
            self.n_cells = self._lib.adaflo_n_cells(self._ctx)
            self.cell_diameter, self.minimal_edge_length = max(mesh.hd), min(mesh.hd)
            return
        d = _lib.BrickDesc()
        d.dim = mesh.dim
        for i in range(3):
            d.ncell[i], d.h[i], d.origin[i] = mesh.ncell[i], mesh.h[i], mesh.lower[i]
        d.velocity_degree, d.ls_degree = velocity_degree, ls_degree
        d.ls_constrained = sum(1 << f for f in constrained_faces)
        d.velocity_constrained = sum(1 << (3 * f + c) for f in dirichlet_faces_u for c in range(3))
        d.device, d.stream = device, None
        ctx = _lib.CtxHandle()
        code = self._lib.adaflo_ctx_create(C.byref(d), C.byref(ctx))
        if code != 0:
            raise _lib.AdafloError("adaflo_ctx_create failed (%d): %s" % (
                code, self._lib.adaflo_last_error(None).decode()))
        ctx.alive = True
        self._ctx = ctx
        if stream is not None:
            _lib.check(ctx, self._lib.adaflo_set_stream(ctx, stream or None))
        self.n_dofs = self._lib.adaflo_n_dofs_ls(ctx)
        self.n_q = self._lib.adaflo_n_q_points_ls(ctx)
        self.n_cells = self._lib.adaflo_n_cells(ctx)
        # compute_cell_diameters on a Cartesian mesh, include/adaflo/util.h:47-120
        self.cell_diameter = max(mesh.hd)
        self.minimal_edge_length = min(mesh.hd)

    def __del__(self):
        try:
            if self._ctx is not None and self._owns_ctx:
                self._ctx.alive = False
                self._lib.adaflo_ctx_destroy(self._ctx)
            self._ctx = None
        except Exception:
            pass

    def vector(self, values=None, blocks=1):
        v = DeviceVector(self._ctx, self.n_dofs * blocks)
        v.set(np.zeros(v.n) if values is None else values)
        return v

    def velocity_vector(self, values):
        return DeviceVector.from_numpy(self._ctx, values)

    def set_parameters(self, epsilon_used, time_step, weight=1.0, weight_old=-1.0, weight_old_old=0.0,
                       epsilon=1.0):
        p = _lib.LSParams(epsilon_used, self.minimal_edge_length, time_step, weight, weight_old,
                          weight_old_old, epsilon)
        _lib.check(self._ctx, self._lib.adaflo_ls_set_params(self._ctx, C.byref(p)))

    def compute_heaviside(self, heaviside, level_set, epsilon):
        """LevelSetOKZSolver::compute_heaviside, level_set_okz.cc:479-540 (epsilon = parameters.epsilon)"""
        _lib.check(self._ctx, self._lib.adaflo_ls_compute_heaviside(self._ctx, heaviside.ptr, level_set.ptr, epsilon))

    def compute_force(self, user_rhs_u, heaviside, curvature, parameters):
        """the cell loop of LevelSetOKZSolver::compute_force, level_set_okz.cc:317-432: adds the
        surface-tension + gravity force to user_rhs_u (zero it first, :420) and refreshes the
        variable density / viscosity arrays of the engine context"""
        p = _lib.ForceParams(parameters.surface_tension, parameters.gravity, parameters.density,
                             parameters.density_diff, parameters.viscosity, parameters.viscosity_diff,
                             int(parameters.interpolate_grad_onto_pressure))
        _lib.check(self._ctx, self._lib.adaflo_ls_compute_force(self._ctx, user_rhs_u.ptr, heaviside.ptr,
                                                                curvature.ptr, C.byref(p)))

    def initialize_mass_matrix_diagonal(self):
        """initialize_mass_matrix_diagonal (level_set_okz_preconditioner.h:35-76): the
        DiagonalPreconditioner of all level-set solves; also registered for the constrained rows"""
        from .solvers import DiagonalPreconditioner
        diag = self.vector()
        _lib.check(self._ctx, self._lib.adaflo_ls_mass_matrix_diagonal(self._ctx, diag.ptr))
        self.set_diagonal(diag)
        return DiagonalPreconditioner(diag)

    def set_kernel_variant(self, variant):
        """0: generic per-cell kernels, 1 (default): structured Q1 sweep kernel for the operator
        applications (FE_Q_iso_Q1(s) = trilinear elements on the s-times refined grid)"""
        _lib.check(self._ctx, self._lib.adaflo_set_kernel_variant(self._ctx, variant))

    def set_diagonal(self, diag):
        """DiagonalPreconditioner::get_vector() used on constrained rows"""
        _lib.check(self._ctx, self._lib.adaflo_ls_set_diagonal(self._ctx, diag.ptr))

    def _q(self, getter):
        out = np.empty(self.n_cells * self.n_q * 3)
        _lib.check(self._ctx, getter(self._ctx, out.ctypes.data, 0))
        return out

    def _set_q(self, setter, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.size == self.n_cells * self.n_q * 3
        _lib.check(self._ctx, setter(self._ctx, a.ctypes.data, 0))


class LevelSetOKZSolverAdvanceConcentration:
    def __init__(self, ops):
        self.ops, self._lib, self._ctx = ops, ops._lib, ops._ctx

    def advance_concentration_vmult(self, dst, src):
        _lib.check(self._ctx, self._lib.adaflo_ls_advance_concentration_vmult(self._ctx, dst.ptr, src.ptr))

    def local_advance_concentration_rhs(self, dst, solution, solution_old, solution_old_old,
                                        vel_solution, use_old_old):
        _lib.check(self._ctx, self._lib.adaflo_ls_advance_concentration_rhs(
            self._ctx, dst.ptr, solution.ptr, solution_old.ptr, solution_old_old.ptr,
            vel_solution.ptr, int(use_old_old)))

    # ---- parameters.convection_stabilization (level_set_okz_advance_concentration.cc:248-249,
    # ---- 344-369, 387-388, 419-472, 569-617)
    def set_convection_stabilization(self, enabled, symmetry_faces=()):
        """global_omega_diameter = diameter_on_coarse_grid: the space diagonal of the brick"""
        m = self.ops.mesh
        diameter = float(np.sqrt(sum((u - l) ** 2 for u, l in zip(m.upper[:m.dim], m.lower[:m.dim]))))
        mask = sum(1 << f for f in symmetry_faces)
        _lib.check(self._ctx, self._lib.adaflo_ls_set_convection_stabilization(self._ctx, int(enabled), diameter, mask))
        self.convection_stabilization = bool(enabled)
        self.global_omega_diameter = diameter

    def get_maximal_velocity(self, vel_solution):
        r = C.c_double()
        _lib.check(self._ctx, self._lib.adaflo_ls_max_velocity(self._ctx, vel_solution.ptr, C.byref(r)))
        return r.value

    def local_advance_concentration_rhs_stabilized(self, dst, solution, solution_old, solution_old_old, vel_solution,
                                                   vel_solution_old, vel_solution_old_old, use_old_old, old_step_size,
                                                   global_max_velocity):
        """right-hand side with the artificial viscosities of this step (written to the public array)
        and the boundary part the reference's driver adds afterwards"""
        _lib.check(self._ctx, self._lib.adaflo_ls_advance_concentration_rhs_stabilized(
            self._ctx, dst.ptr, solution.ptr, solution_old.ptr, solution_old_old.ptr, vel_solution.ptr,
            vel_solution_old.ptr, vel_solution_old_old.ptr, int(use_old_old), float(old_step_size),
            float(global_max_velocity)))

    @property
    def artificial_viscosities(self):
        out = np.empty(self.ops.n_cells)
        _lib.check(self._ctx, self._lib.adaflo_ls_get_artificial_viscosities(self._ctx, out.ctypes.data, 0))
        return out

    @artificial_viscosities.setter
    def artificial_viscosities(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.size == self.ops.n_cells
        _lib.check(self._ctx, self._lib.adaflo_ls_set_artificial_viscosities(self._ctx, a.ctypes.data, 0))

    def advance_concentration(self, solution, solution_old, solution_old_old, vel_solution, rhs, increment,
                              preconditioner, use_old_old=True, tol_nl_iteration=1e-8, vel_solution_old=None,
                              vel_solution_old_old=None, old_step_size=None):
        """LevelSetOKZSolverAdvanceConcentration::advance_concentration
        (level_set_okz_advance_concentration.cc:549-660): right-hand side, BiCGStab
        with ReductionControl(30, 0.05 tol_nl, 1e-8), solution += increment.  The caller has advanced
        the level-set TimeStepping and pushed its weights (set_parameters).  With
        set_convection_stabilization(True) the old velocities and the old step size are needed for the
        artificial viscosities (:344-369).  Returns (iterations, initial residual) like the reference
        prints them."""
        from .solvers import AdvanceConcentrationMatrix, NoConvergence, ReductionControl, SolverBicgstab
        rhs.fill(0.0)
        if getattr(self, "convection_stabilization", False):
            vmax = self.get_maximal_velocity(vel_solution)          # :548-551
            self.local_advance_concentration_rhs_stabilized(rhs, solution, solution_old, solution_old_old, vel_solution,
                                                            vel_solution_old, vel_solution_old_old, use_old_old,
                                                            old_step_size, vmax)
        else:
            self.local_advance_concentration_rhs(rhs, solution, solution_old, solution_old_old, vel_solution,
                                                 use_old_old)
        control = ReductionControl(30, 0.05 * tol_nl_iteration, 1e-8)
        increment.fill(0.0)
        try:
            SolverBicgstab(control).solve(AdvanceConcentrationMatrix(self.ops), increment, rhs, preconditioner)
        except NoConvergence:
            # the reference falls back to GMRES here (:634-641); retry with a longer BiCGStab run
            control = ReductionControl(3000, 0.05 * tol_nl_iteration, 1e-8)
            increment.fill(0.0)
            SolverBicgstab(control).solve(AdvanceConcentrationMatrix(self.ops), increment, rhs, preconditioner)
        solution.add(increment)
        return control.last_step(), control.initial_value()

    @property
    def evaluated_convection(self):
        return self.ops._q(self._lib.adaflo_ls_get_evaluated_convection)

    @evaluated_convection.setter
    def evaluated_convection(self, a):
        self.ops._set_q(self._lib.adaflo_ls_set_evaluated_convection, a)


class LevelSetOKZSolverReinitialization:
    def __init__(self, ops):
        self.ops, self._lib, self._ctx = ops, ops._lib, ops._ctx

    def reinitialization_vmult(self, dst, src, diffuse_only):
        _lib.check(self._ctx, self._lib.adaflo_ls_reinitialization_vmult(self._ctx, dst.ptr, src.ptr,
                                                                        int(diffuse_only)))

    def local_reinitialize_rhs(self, dst, solution, normal_vector_field, diffuse_only, first_reinit_step):
        nptr = normal_vector_field.ptr if normal_vector_field is not None else None
        _lib.check(self._ctx, self._lib.adaflo_ls_reinitialization_rhs(
            self._ctx, dst.ptr, solution.ptr, nptr, int(diffuse_only), int(first_reinit_step)))

    def reinitialize(self, solution, normal_vector_field, rhs, increment, preconditioner, stab_steps,
                     diff_steps=0, compute_normal=None, last_concentration_range=(-1.0, 1.0)):
        """LevelSetOKZSolverReinitialization::reinitialize (level_set_okz_reinitialization.cc:255-375):
        diff_steps diffusion-only steps, then stab_steps Olsson-Kreiss-Zahedi steps; the normal is
        recomputed (compute_normal(True)) before the first of them.  Returns the CG iteration counts."""
        from .solvers import ReductionControl, ReinitializationMatrix, SolverCG
        actual_diff_steps = diff_steps
        if last_concentration_range[0] < -1.02 or last_concentration_range[1] > 1.02:
            actual_diff_steps += 3
        iterations = []
        for tau in range(actual_diff_steps + stab_steps):
            first_reinit_step = tau == actual_diff_steps
            if first_reinit_step and compute_normal is not None:
                compute_normal(True)
            diffuse = tau < actual_diff_steps
            rhs.fill(0.0)
            self.local_reinitialize_rhs(rhs, solution, normal_vector_field, diffuse, first_reinit_step)
            increment.fill(0.0)
            control = ReductionControl(2000, 1e-50, 1e-6)
            SolverCG(control).solve(ReinitializationMatrix(self.ops, diffuse), increment, rhs, preconditioner)
            iterations.append(control.last_step())
            solution.add(increment)
            if increment.l2_norm() < 1e-6:
                break
        return iterations

    @property
    def evaluated_normal(self):
        return self.ops._q(self._lib.adaflo_ls_get_evaluated_normal)

    @evaluated_normal.setter
    def evaluated_normal(self, a):
        self.ops._set_q(self._lib.adaflo_ls_set_evaluated_normal, a)


class LevelSetOKZSolverComputeNormal:
    def __init__(self, ops):
        self.ops, self._lib, self._ctx = ops, ops._lib, ops._ctx

    def compute_normal_vmult(self, dst, src):
        _lib.check(self._ctx, self._lib.adaflo_ls_compute_normal_vmult(self._ctx, dst.ptr, src.ptr))

    def local_compute_normal_rhs(self, dst, level_set_solution):
        _lib.check(self._ctx, self._lib.adaflo_ls_compute_normal_rhs(self._ctx, dst.ptr, level_set_solution.ptr))

    def compute_normal(self, normal_vector_field, normal_vector_rhs, level_set_solution, preconditioner,
                       fast_computation=False):
        """LevelSetOKZSolverComputeNormal::compute_normal (level_set_okz_compute_normal.cc:207-285) in
        its matrix-free form (:262: solver.solve(matrix, field, rhs, preconditioner); the production
        code solves the same system with an assembled matrix + ILU): projection of grad(phi), CG to
        1e-7 (1e-5 fast), starting from the previous normal field"""
        from .solvers import ComputeNormalMatrix, ReductionControl, SolverCG
        normal_vector_rhs.fill(0.0)
        self.local_compute_normal_rhs(normal_vector_rhs, level_set_solution)
        if self.ops.exact_projection and _projection_solve(self, normal_vector_field, normal_vector_rhs, 3):
            # the projection matrix is a constant-coefficient tensor-product operator on the brick: solved
            # exactly by fast diagonalisation (csrc/fdm.hip), 0 iterations
            return 0
        control = ReductionControl(4000, 1e-50, 1e-5 if fast_computation else 1e-7)
        SolverCG(control).solve(ComputeNormalMatrix(self.ops), normal_vector_field, normal_vector_rhs, preconditioner)
        return control.last_step()


class LevelSetOKZSolverComputeCurvature:
    def __init__(self, ops):
        self.ops, self._lib, self._ctx = ops, ops._lib, ops._ctx

    def compute_curvature_vmult(self, dst, src, apply_diffusion):
        _lib.check(self._ctx, self._lib.adaflo_ls_compute_curvature_vmult(self._ctx, dst.ptr, src.ptr,
                                                                         int(apply_diffusion)))

    def local_compute_curvature_rhs(self, dst, normal_vector_field):
        _lib.check(self._ctx, self._lib.adaflo_ls_compute_curvature_rhs(self._ctx, dst.ptr, normal_vector_field.ptr))

    def compute_curvature(self, solution_curvature, rhs, normal_vector_field, preconditioner, solution_ls=None,
                          use_projection_matrix=True):
        """LevelSetOKZSolverComputeCurvature::compute_curvature (level_set_okz_compute_curvature.cc:
        325-376): projection of -div(n), CG to 1e-8 starting from the previous curvature.  Like the
        reference (:355) the system matrix is the projection matrix shared with the normal solve
        (mass + 4 delta Laplace, adaflo_ls_projection_vmult); use_projection_matrix=False solves with
        ComputeCurvatureMatrix instead (the call the reference keeps commented out at :354).  With
        solution_ls given, followed by the curvature correction (:360-376,
        parameters.curvature_correction)"""
        from .solvers import ComputeCurvatureMatrix, ProjectionMatrix, ReductionControl, SolverCG
        rhs.fill(0.0)
        self.local_compute_curvature_rhs(rhs, normal_vector_field)
        control = ReductionControl(2000, 1e-50, 1e-8)
        if not (self.ops.exact_projection and use_projection_matrix
                and _projection_solve(self, solution_curvature, rhs, 1)):
            matrix = ProjectionMatrix(self.ops) if use_projection_matrix else ComputeCurvatureMatrix(self.ops)
            SolverCG(control).solve(matrix, solution_curvature, rhs, preconditioner)
        if solution_ls is not None:
            ops = self.ops
            _lib.check(ops._ctx, ops._lib.adaflo_ls_curvature_correction(ops._ctx, solution_curvature.ptr, solution_ls.ptr))
        return control.last_step()

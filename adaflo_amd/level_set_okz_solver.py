"""Host-side mirror of adaflo::LevelSetOKZSolver (two-phase flow with the conservative level set
method of Olsson, Kreiss and Zahedi) on a uniform brick, every vector and every kernel on the GPU:

    LevelSetBaseAlgorithm::advance_time_step     source/level_set_base.cc:190-291
    TwoPhaseBaseAlgorithm::init_time_advance     source/two_phase_base.cc:441-475
    LevelSetOKZSolver::advance_concentration /
      reinitialize / compute_normal / compute_curvature / compute_heaviside / compute_force
                                                 source/level_set_okz.cc:317-560

One engine context carries the Navier-Stokes operator and the level-set operators, so that
compute_force writes the density / viscosity arrays the two-phase Jacobian reads.  Not mirrored:
adaptive mesh refinement (hence no `last_refine_step`), output."""
import numpy as np

from . import level_set_okz as lso
from .navier_stokes import NavierStokes, node_coordinates
from .time_stepping import TimeStepping


class LevelSetOKZSolver:
    def __init__(self, parameters, mesh, distance_function, dirichlet_function=None, device=0,
                 n_reinit_steps=2, n_initial_reinit_steps=2, exact_projection=None, symmetry_faces=()):
        """distance_function(xyz[n][3]) -> signed distance to the interface, positive outside the
        second fluid (tests/rising_bubble.cc:59-77).  exact_projection: solve the normal / curvature
        projections (the reference: CG + ILU on the assembled projection matrix to 1e-7 / 1e-8) exactly by
        fast diagonalisation instead of by diagonally preconditioned CG to those tolerances (default: where the engine
        has the inverses, i.e. for dim = 3); symmetry_faces: see NavierStokes"""
        p = parameters
        self.parameters, self.mesh = p, mesh
        s, k = p.concentration_subdivisions, p.velocity_degree
        if dirichlet_function is None:
            dirichlet_function = lambda x, t: np.zeros_like(x)          # no-slip box
        self.time_stepping = TimeStepping(p)
        self.navier_stokes = NavierStokes(p, mesh, self.time_stepping, dirichlet_function, device=device, ls_degree=s,
                                          symmetry_faces=symmetry_faces)
        self.ops = lso.LevelSetOperators(mesh, s, velocity_degree=k,
                                         navier_stokes_matrix=self.navier_stokes.navier_stokes_matrix)
        if exact_projection is None:            # default: exact where the engine can (the inverses are dim = 3 only)
            exact_projection = mesh.dim == 3
        self.ops.exact_projection = bool(exact_projection)
        # two_phase_base.cc:282-291: epsilon_used = epsilon / subdivisions * largest edge length
        self.epsilon_used = p.epsilon / s * max(mesh.hd)
        # the sub-operators advance their own TimeStepping copies (level_set_okz.cc:94-106)
        self.ts_advect, self.ts_reinit = TimeStepping(p), TimeStepping(p)
        self.advection_operator = lso.LevelSetOKZSolverAdvanceConcentration(self.ops)
        if p.convection_stabilization:
            self.advection_operator.set_convection_stabilization(True)
        self.reinit_operator = lso.LevelSetOKZSolverReinitialization(self.ops)
        self.normal_operator = lso.LevelSetOKZSolverComputeNormal(self.ops)
        self.curvature_operator = lso.LevelSetOKZSolverComputeCurvature(self.ops)
        self.n_reinit_steps = n_reinit_steps
        # level_set_base.cc:61-63: bookkeeping of the "correct excessive residual" branch;
        # two_phase_base.h:233: range of the level set seen by the last get_concentration_range()
        self.old_residual = float("inf")
        self.last_smoothing_step = 0
        self.last_concentration_range = (0.0, 0.0)
        self.smoothing_steps = []
        v = self.ops.vector
        self.solution, self.solution_old, self.solution_old_old = v(), v(), v()     # block 0: level set
        self.curvature, self.curvature_old, self.curvature_old_old = v(), v(), v()  # block 1: curvature
        self.heaviside, self.system_rhs, self.solution_update, self.tmp = v(), v(), v(), v()
        self.normal_vector_field, self.normal_vector_rhs = v(blocks=3), v(blocks=3)
        self._push_ls_parameters(self.time_stepping)
        self.preconditioner = self.ops.initialize_mass_matrix_diagonal()
        self._x_ls = self._ls_node_coordinates()
        # initial profile (level_set_okz.cc:203-210) and initial reinitialisation steps
        dist = np.asarray(distance_function(self._x_ls), dtype=np.float64)
        self.solution.set(-np.tanh(dist / (2.0 * self.epsilon_used)))
        self.reinit_iterations = []
        if n_initial_reinit_steps > 0:
            self.reinitialize(n_initial_reinit_steps)
        self.initial_reinit_iterations = self.reinit_iterations.pop() if self.reinit_iterations else []
        self.solution_old.sadd(0.0, 1.0, self.solution)
        self.solution_old_old.sadd(0.0, 1.0, self.solution)
        self.concentration_iterations = []

    # ------------------------------------------------------------------------------------------
    def _ls_node_coordinates(self):
        m, s = self.mesh, self.parameters.concentration_subdivisions
        ax = [np.linspace(m.lower[d], m.upper[d], s * m.ncell[d] + 1) if d < m.dim else np.zeros(1) for d in range(3)]
        z, y, x = np.meshgrid(ax[2], ax[1], ax[0], indexing="ij")
        return np.stack([x.reshape(-1), y.reshape(-1), z.reshape(-1)], axis=1)

    def _push_ls_parameters(self, ts):
        self.ops.set_parameters(self.epsilon_used, ts.step_size(), ts.weight(), ts.weight_old(), ts.weight_old_old(),
                                self.parameters.epsilon)

    # ---- the pieces of LevelSetBaseAlgorithm::advance_time_step ---------------------------------
    def init_time_advance(self):
        """two_phase_base.cc:441-460: Navier-Stokes first, then extrapolate / shift the level set"""
        self.navier_stokes.init_time_advance()
        ts = self.time_stepping
        step, old = ts.step_size(), ts.old_step_size()
        for cur, o, oo in ((self.solution, self.solution_old, self.solution_old_old),
                           (self.curvature, self.curvature_old, self.curvature_old_old)):
            self.tmp.sadd(0.0, 1.0, cur)
            if old > 0:
                self.tmp.sadd((step + old) / old, -step / old, o)
            oo.sadd(0.0, 1.0, o)
            o.sadd(0.0, 1.0, cur)
            cur.sadd(0.0, 1.0, self.tmp)

    def advance_concentration(self):
        ts = self.ts_advect
        ts.set_desired_time_step(self.time_stepping.step_size())     # advance_concentration.cc:508
        ts.next()
        self._push_ls_parameters(ts)
        vel = self.navier_stokes.navier_stokes_matrix.wrap(self.navier_stokes.solution[0])
        use_old_old = ts.scheme == "bdf_2" and ts.step_no() > 1          # advance_concentration.cc:375-378
        ns, wrap = self.navier_stokes, self.navier_stokes.navier_stokes_matrix.wrap
        it = self.advection_operator.advance_concentration(
            self.solution, self.solution_old, self.solution_old_old, vel, self.system_rhs, self.solution_update,
            self.preconditioner, use_old_old, self.parameters.tol_nl_iteration,
            vel_solution_old=wrap(ns.solution_old[0]), vel_solution_old_old=wrap(ns.solution_old_old[0]),
            old_step_size=ts.old_step_size() if ts.old_step_size() > 0 else ts.step_size())
        self.concentration_iterations.append(it)

    def compute_normal(self, fast_computation):
        return self.normal_operator.compute_normal(self.normal_vector_field, self.normal_vector_rhs, self.solution,
                                                   self.preconditioner, fast_computation)

    def reinitialize(self, stab_steps, diff_steps=0):
        ts = self.ts_reinit
        ts.set_desired_time_step(self.time_stepping.step_size())     # reinitialization.cc:264
        self._push_ls_parameters(ts)
        its = self.reinit_operator.reinitialize(self.solution, self.normal_vector_field, self.system_rhs,
                                                self.solution_update, self.preconditioner, stab_steps, diff_steps,
                                                compute_normal=self.compute_normal,
                                                last_concentration_range=self.last_concentration_range)
        ts.next()
        self.reinit_iterations.append(its)

    def compute_curvature(self):
        self.compute_normal(False)
        return self.curvature_operator.compute_curvature(
            self.curvature, self.system_rhs, self.normal_vector_field, self.preconditioner,
            solution_ls=self.solution if self.parameters.curvature_correction else None)

    def compute_force(self):
        """level_set_okz.cc:415-432"""
        self.ops.compute_heaviside(self.heaviside, self.solution, self.parameters.epsilon)
        self.compute_curvature()
        ns = self.navier_stokes
        ns.user_rhs[0].zero_()
        ns.user_rhs[1].zero_()
        self.ops.compute_force(ns.navier_stokes_matrix.wrap(ns.user_rhs[0]), self.heaviside, self.curvature,
                               self.parameters)

    def advance_time_step(self):
        """level_set_base.cc:248-291 (do_iteration = false).  When the curvature gets bad the initial
        Navier-Stokes residual jumps: if it is at least twice that of the previous step (and the last
        smoothing is more than three steps ago) ten additional diffusion steps are taken, the force is
        recomputed and the residual evaluated again (:262-278)"""
        self.init_time_advance()
        self.advance_concentration()
        self.reinitialize(self.n_reinit_steps)
        self.compute_force()
        ns = self.navier_stokes
        actual_res = ns.compute_initial_residual()
        step_no = self.time_stepping.step_no()
        if step_no > 3 + self.last_smoothing_step and actual_res >= 2.0 * self.old_residual:
            self.reinitialize(self.n_reinit_steps, 10)
            self.compute_force()
            actual_res = ns.compute_initial_residual()
            self.last_smoothing_step = step_no
            self.smoothing_steps.append(step_no)
        self.old_residual = actual_res
        return ns.solve_nonlinear_system(actual_res)

    def get_concentration_range(self):
        """TwoPhaseBaseAlgorithm::get_concentration_range (two_phase_base.cc:515-545): smallest / largest
        value of the level set on the (s + 2)-times iterated trapezoid points of every cell; remembered
        for the next reinitialize(), which adds three diffusion steps once the profile has left
        [-1.02, 1.02] (reinitialization.cc:281-284)"""
        m, s = self.mesh, self.parameters.concentration_subdivisions
        nn = m.nodes(s)
        phi = self.solution.numpy().reshape(nn[2], nn[1], nn[0])
        t = np.arange(s + 3) / (s + 2.0) * s                 # positions in units of sub-cells
        i0 = np.minimum(t.astype(int), s - 1)
        w = np.zeros((s + 3, s + 1))
        w[np.arange(s + 3), i0] = 1.0 - (t - i0)
        w[np.arange(s + 3), i0 + 1] = t - i0
        lo, hi = np.inf, -np.inf
        for cz in range(m.ncell[2]):                          # one layer of cells at a time (memory)
            if m.dim == 2:
                blk = phi                                   # one node layer: nothing to interpolate in z
            else:
                blk = phi[s * cz:s * cz + s + 1]
                blk = np.einsum("pk,kyx->pyx", w, blk)
            ys = np.lib.stride_tricks.sliding_window_view(blk, s + 1, axis=1)[:, ::s]       # [p][cy][x][j]
            blk = np.einsum("qj,pcxj->pcqx", w, ys)
            xs = np.lib.stride_tricks.sliding_window_view(blk, s + 1, axis=3)[:, :, :, ::s]  # [p][cy][q][cx][i]
            val = np.einsum("ri,pcqxi->pcqxr", w, xs)
            lo, hi = min(lo, float(val.min())), max(hi, float(val.max()))
        self.last_concentration_range = (lo, hi)
        return self.last_concentration_range

    # ---- diagnostics (tests/rising_bubble.cc evaluates the same quantities) ----------------------
    def compute_bubble_statistics(self):
        """TwoPhaseBaseAlgorithm<2>::compute_bubble_statistics (two_phase_base.cc:621-905; dim = 2 only, as the
        reference's tests use it): dict(area, perimeter, circularity, velocity, centre) and the three printed lines"""
        from .two_phase_statistics import compute_bubble_statistics, format_bubble_statistics
        p, m = self.parameters, self.mesh
        stat = compute_bubble_statistics(m, p.concentration_subdivisions, p.velocity_degree, self.solution.numpy(),
                                         self.navier_stokes.solution[0].cpu().numpy().reshape(-1, 3))
        diameter = float(np.sqrt(sum((u - l) ** 2 for u, l in zip(m.upper[:m.dim], m.lower[:m.dim]))))
        stat["lines"] = format_bubble_statistics(stat, diameter)
        return stat


    def bubble_volume_and_centre(self):
        """integral of H and centre of mass of the second fluid (lumped with the mass diagonal)"""
        w = self.preconditioner.diagonal_vector.numpy()          # (phi_i, phi_i); proportional to the lumped weight
        h = self.heaviside.numpy()
        vol = float(w @ h)
        return vol, (self._x_ls * (w * h)[:, None]).sum(axis=0) / vol

"""Host-side mirror of adaflo::NavierStokes<dim> (source/navier_stokes.cc) for a uniform brick:
the time loop and the Newton iteration around the operators, with every vector resident in HBM.

    init_time_advance        :659-745   TimeStepping::next, shift of the old solutions, extrapolated
                                        initial guess, Dirichlet values of the new time level
    compute_residual         :781-800   residual + mean-value projection of the pressure rows
    solve_nonlinear_system   :832-960   Newton with the linear tolerance rule of :862-876
    solve_system             :561-653   FGMRES(50) + NavierStokesPreconditioner with inner solves
                                        (native: csrc/krylov.hip, adaflo_ns_solve_system)
    advance_time_step        :964-990

Vectors are torch CUDA tensors (device memory + the small vector algebra of the driver) wrapped
as DeviceVectors for the engine; the engine runs on torch's current stream."""
import ctypes as C

import numpy as np

from . import _lib
from .navier_stokes_matrix import NavierStokesMatrix
from .vectors import BlockVector, DeviceVector


def gauss_lobatto_points(n):
    """n Gauss-Lobatto points on [0, 1] (support points of FE_Q(QGaussLobatto(n)), navier_stokes.cc:95)"""
    if n == 2:
        return np.array([0.0, 1.0])
    c = np.zeros(n)
    c[-1] = 1.0                                   # P_{n-1}
    inner = np.polynomial.legendre.Legendre(c).deriv().roots()
    return 0.5 * (np.concatenate([[-1.0], np.sort(inner.real), [1.0]]) + 1.0)


def node_coordinates(mesh, degree):
    """coordinates [n_nodes][3] of the lexicographic nodes of FE_Q(degree) on the brick (dim = 2: z = 0)"""
    axes = []
    gl = gauss_lobatto_points(degree + 1)
    for d in range(3):
        if d >= mesh.dim:
            axes.append(np.zeros(1))
            continue
        h = mesh.h[d]
        x = np.concatenate([[mesh.lower[d]]] + [mesh.lower[d] + h * (c + gl[1:]) for c in range(mesh.ncell[d])])
        axes.append(x)
    z, y, x = np.meshgrid(axes[2], axes[1], axes[0], indexing="ij")
    return np.stack([x.reshape(-1), y.reshape(-1), z.reshape(-1)], axis=1)


def open_boundary_rhs(mesh, k, open_faces, t):
    """const_rhs of NavierStokes::apply_boundary_conditions (source/navier_stokes.cc:1259-1310): for every open face
    int_face (phi_i . n) p_ext dS with QGauss(k + 1) per face direction, p_ext(x[n][3], t) the prescribed pressure.
    Returns [n_nodes][3] (constrained rows are NOT removed here)."""
    nn = mesh.nodes(k)
    rhs = np.zeros((nn[2], nn[1], nn[0], 3))
    gl = gauss_lobatto_points(k + 1)
    xg, wg = np.polynomial.legendre.leggauss(k + 1)
    xg, wg = 0.5 * (xg + 1.0), 0.5 * wg
    S = np.ones((len(xg), k + 1))                                # S[q][i]: Lagrange basis through the GL points
    for i in range(k + 1):
        for j in range(k + 1):
            if i != j:
                S[:, i] *= (xg - gl[j]) / (gl[i] - gl[j])
    for f, pressure in open_faces.items():
        d, side = f // 2, f % 2
        normal = 1.0 if side else -1.0
        other = [e for e in range(mesh.dim) if e != d]
        # quadrature points of all face cells: coordinate arrays per remaining direction [cell][q]
        coords = [mesh.lower[e] + mesh.h[e] * (np.arange(mesh.ncell[e])[:, None] + xg[None, :]) for e in other]
        x = np.zeros([len(c.reshape(-1)) for c in coords] + [3])
        x[..., d] = mesh.upper[d] if side else mesh.lower[d]
        if len(other) == 0:                                  # dim = 1: the "face integral" is the point value
            node = (nn[0] - 1) if side else 0
            rhs.reshape(-1, 3)[node, 0] += normal * float(np.asarray(pressure(x.reshape(-1, 3), t)).reshape(-1)[0])
            continue
        if len(other) == 1:
            x[:, other[0]] = coords[0].reshape(-1)
        else:
            x[:, :, other[0]] = coords[0].reshape(-1)[:, None]
            x[:, :, other[1]] = coords[1].reshape(-1)[None, :]
        pq = np.asarray(pressure(x.reshape(-1, 3), t), dtype=np.float64)
        idx = [np.arange(mesh.ncell[e])[:, None] * k + np.arange(k + 1)[None, :] for e in other]   # [cell][i]
        plane = [slice(None)] * 3
        plane[2 - d] = nn[d] - 1 if side else 0
        if len(other) == 1:
            e = other[0]
            pq = pq.reshape(mesh.ncell[e], k + 1)
            loc = normal * mesh.h[e] * np.einsum("qi,cq,q->ci", S, pq, wg)
            line = np.zeros(nn[e])
            np.add.at(line, idx[0], loc)
            target = rhs[tuple(plane)]                       # the two remaining axes (z, e) -> flat dim: z has one node
            target.reshape(-1, 3)[:, d] += line
        else:
            e0, e1 = other                                   # e0 < e1: x[...] axes are (cells of e0, cells of e1)
            pq = pq.reshape(mesh.ncell[e0], k + 1, mesh.ncell[e1], k + 1)
            loc = normal * mesh.h[e0] * mesh.h[e1] * np.einsum("qi,rj,aqbr,q,r->aibj", S, S, pq, wg, wg)
            face = np.zeros((nn[e0], nn[e1]))
            np.add.at(face, (idx[0][:, :, None, None], idx[1][None, None, :, :]), loc)
            target = rhs[tuple(plane)]                       # axes in (slower, faster) memory order = (e1, e0)
            target[..., d] += face.T
    return rhs.reshape(-1, 3)


class NavierStokes:
    def __init__(self, parameters, mesh, time_stepping, dirichlet_function, device=0, ls_degree=0, symmetry_faces=(),
                 open_faces=None):
        """dirichlet_function(xyz[n][3], t) -> velocity[n][3] (dim = 2: [n][2] is fine) on the Dirichlet boundary = every
        face not in symmetry_faces; on those only the normal component is constrained, to zero
        (FlowBaseAlgorithm::set_symmetry_boundary; tests/rising_bubble.cc:133-150 uses it for the side walls);
        open_faces {face: p_ext(x[n][3], t)}: open boundaries with normal flux (set_open_boundary_with_normal_flux,
        tests/poiseuille.cc:252-255): tangential components constrained to zero, the prescribed pressure enters the
        residual as a face integral, and the pressure level is fixed by it (no mean-value projection);
        ls_degree > 0 adds the level-set spaces to the engine context (two-phase flow)"""
        import torch
        if parameters.linearization == "projection":
            # the pressure-projection scheme needs the p^n swap and the phi extrapolation of
            # navier_stokes.cc:689-717,839-841; the operators support it, this driver does not
            raise NotImplementedError("linearization = projection is not driven by NavierStokes (operators only)")
        self.parameters, self.mesh, self.time_stepping = parameters, mesh, time_stepping
        self.dirichlet_function = dirichlet_function
        self.device = torch.device("cuda", device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        symmetry_faces = sorted(set(int(f) for f in symmetry_faces))
        self.open_faces = dict(open_faces or {})
        assert all(0 <= f < 2 * mesh.dim for f in symmetry_faces) and not set(symmetry_faces) & set(self.open_faces)
        dirichlet_faces = [f for f in range(2 * mesh.dim) if f not in symmetry_faces and f not in self.open_faces]
        self.navier_stokes_matrix = NavierStokesMatrix(parameters, mesh, dirichlet_faces_u=dirichlet_faces,
                                                       symmetry_faces_u=symmetry_faces,
                                                       normal_flux_faces_u=sorted(self.open_faces),
                                                       device=device, stream=stream, ls_degree=ls_degree)
        # constant pressure modes only without an open boundary (navier_stokes.cc:330-346)
        self.navier_stokes_matrix.initialize(time_stepping, not self.open_faces)
        m = self.navier_stokes_matrix
        k = parameters.velocity_degree
        self._lib, self._ctx = _lib.load(), m._require()
        # boundary nodes and their coordinates (apply_boundary_conditions :1216-1257)
        nn = mesh.nodes(k)
        idx = np.indices((nn[2], nn[1], nn[0]))
        on_face = lambda f: (idx[2 - f // 2] == (nn[f // 2] - 1 if f % 2 else 0)).reshape(-1)
        on_b = np.zeros(mesh.n_nodes(k), dtype=bool)
        for f in dirichlet_faces:
            on_b |= on_face(f)
        self._bnodes = np.nonzero(on_b)[0]
        self._bxyz = node_coordinates(mesh, k)[self._bnodes]
        dof = (3 * self._bnodes[:, None] + np.arange(3)[None, :]).reshape(-1)
        self._bdofs = torch.from_numpy(dof).to(self.device)
        sym = [3 * np.nonzero(on_face(f) & ~on_b)[0] + f // 2 for f in symmetry_faces]
        for f in self.open_faces:                                # tangential components on the open faces
            sym += [3 * np.nonzero(on_face(f) & ~on_b)[0] + c for c in range(mesh.dim) if c != f // 2]
        self._symdofs = torch.from_numpy(np.concatenate(sym) if sym else np.zeros(0, dtype=np.int64)).to(self.device)
        self.const_rhs_u = None                                  # open-boundary face integrals of the time level
        mk = lambda n: torch.zeros(n, dtype=torch.float64, device=self.device)
        nu, npp = m.n_dofs_u(), m.n_dofs_p()
        self.solution = [mk(nu), mk(npp)]
        self.solution_old = [mk(nu), mk(npp)]
        self.solution_old_old = [mk(nu), mk(npp)]
        self.solution_update = [mk(nu), mk(npp)]
        self.system_rhs = [mk(nu), mk(npp)]
        self.user_rhs = [mk(nu), mk(npp)]     # surface tension + gravity (LevelSetOKZSolver::compute_force)
        # when to rebuild the preconditioner (navier_stokes.cc:121-124, 870-970)
        self.update_preconditioner = True
        self.update_preconditioner_frequency = 0
        self.n_iterations_last_prec_update = 0
        self.time_step_last_prec_update = 0
        self.n_preconditioner_builds = 0
        self.history = []           # (res_u, res_p) per compute_residual, like the reference's table
        self.linear_iterations = []

    def _bv(self, pair):
        m = self.navier_stokes_matrix
        return BlockVector([m.wrap(pair[0]), m.wrap(pair[1])])

    def set_initial_condition(self, u, p):
        import torch
        self.solution[0].copy_(torch.from_numpy(np.ascontiguousarray(u, dtype=np.float64)))
        self.solution[1].copy_(torch.from_numpy(np.ascontiguousarray(p, dtype=np.float64)))

    # ------------------------------------------------------------------------------------------
    def init_time_advance(self):
        import torch
        ts = self.time_stepping
        ts.next()
        self.navier_stokes_matrix.update_parameters()
        for b in range(2):      # :672-686
            cur, old, oo = self.solution[b], self.solution_old[b], self.solution_old_old[b]
            tmp = ts.extrapolate(cur, old)
            oo.copy_(old)
            old.copy_(cur)
            cur.copy_(tmp)
        if len(self._bnodes):
            vals = np.asarray(self.dirichlet_function(self._bxyz, ts.now()), dtype=np.float64).reshape(len(self._bxyz), -1)
            if vals.shape[1] < 3:   # dim < 3: the other components do not exist (constrained to zero on the device)
                vals = np.concatenate([vals, np.zeros((len(vals), 3 - vals.shape[1]))], axis=1)
            vals = np.ascontiguousarray(vals).reshape(-1)
            self.solution[0][self._bdofs] = torch.from_numpy(vals).to(self.device)
        if len(self._symdofs):
            self.solution[0][self._symdofs] = 0.0
        if self.open_faces:                                      # const_rhs, navier_stokes.cc:1259-1310
            rhs = open_boundary_rhs(self.mesh, self.parameters.velocity_degree, self.open_faces, ts.now()).reshape(-1)
            rhs[self._bdofs.cpu().numpy()] = 0.0                 # distribute_local_to_global skips constrained rows
            rhs[self._symdofs.cpu().numpy()] = 0.0
            for c in range(self.mesh.dim, 3):               # components that do not exist
                rhs[c::3] = 0.0
            self.const_rhs_u = torch.from_numpy(rhs).to(self.device)

    def compute_residual(self):
        m = self.navier_stokes_matrix
        # system_rhs.equ(1., const_rhs) (:784; const_rhs = 0 without open boundaries): the residual cell loop accumulates
        if self.const_rhs_u is not None:
            self.system_rhs[0].copy_(self.const_rhs_u)
        else:
            self.system_rhs[0].zero_()
        self.system_rhs[1].zero_()
        m.residual(self._bv(self.system_rhs), self._bv(self.solution), self._bv(self.user_rhs),
                   self._bv(self.solution_old), self._bv(self.solution_old_old))
        m.apply_pressure_average_projection(m.wrap(self.system_rhs[1]))
        res_u, res_p = float(self.system_rhs[0].norm()), float(self.system_rhs[1].norm())
        self.history.append((res_u, res_p))
        return float(np.hypot(res_u, res_p))

    compute_initial_residual = compute_residual     # (:805-827 only prints a table header before)

    cheap_velocity_iterations = 3       # two-phase flow only, see solve_system

    def build_preconditioner(self):
        _lib.check(self._ctx, self._lib.adaflo_ns_preconditioner_setup(self._ctx))
        self.n_preconditioner_builds += 1

    def solve_system(self, linear_tolerance):
        p = self.parameters
        ctl = _lib.SolverControl(p.max_lin_iteration, linear_tolerance, 0.0)
        res = _lib.SolverResult()
        upd, rhs = self.solution_update, self.system_rhs
        _lib.check(self._ctx, self._lib.adaflo_ns_set_iterations_before_inner_solvers(
            self._ctx, int(p.iterations_before_inner_solvers)))
        # variable coefficients (two-phase flow): the cheap first stage of navier_stokes.cc:571-617 applies a velocity
        # solve cut off after a few BiCGStab iterations as approximate inverse (measured, 64 x 64 x 128 cells: 3
        # iterations -> 19-22 outer iterations instead of 15-18 with inner solves to their tolerance, linear solve
        # 0.22-0.26 -> 0.17-0.20 s per time step)
        # (only there: with constant coefficients the first stage is the reference's -- one application of the
        # approximate inverses -- and the header's default 0 stands)
        variable = p.density_diff != 0.0 or p.viscosity_diff != 0.0
        _lib.check(self._ctx, self._lib.adaflo_ns_preconditioner_set_cheap_velocity_iterations(
            self._ctx, self.cheap_velocity_iterations if variable else 0))
        _lib.check(self._ctx, self._lib.adaflo_ns_solve_system(
            self._ctx, upd[0].data_ptr(), upd[1].data_ptr(), rhs[0].data_ptr(), rhs[1].data_ptr(),
            C.byref(ctl), 50, C.byref(res)))
        return res.iterations, res.final_residual

    def solve_nonlinear_system(self, initial_residual):
        """navier_stokes.cc:832-975: at least one linear solve; only the fully implicit schemes iterate
        (the others stop after their one linear system, :906-914); the preconditioner is rebuilt in
        the first step when requested, early inside the iteration when the linear solver needs 1.5 x
        the iterations it needed right after the last rebuild (every sixth step of a stationary
        solve), and requested for the next time step by the iteration-count rules at the end"""
        p, ts = self.parameters, self.time_stepping
        implicit = p.linearization in ("coupled implicit Newton", "coupled implicit Picard")
        res = initial_residual
        n_tot = 0
        premature_update = False
        step = 0
        while step < p.max_nl_iteration:
            linear_tolerance = p.tol_lin_iteration
            if p.rel_lin_iteration:     # :862-876
                if res * p.tol_lin_iteration < 0.5 * p.tol_nl_iteration or not implicit:
                    linear_tolerance = 0.5 * p.tol_nl_iteration
                else:
                    linear_tolerance = min(p.tol_lin_iteration * res, p.tol_lin_iteration)
            if step == 0 and self.update_preconditioner:
                self.build_preconditioner()
            elif ((not premature_update and ts.step_no() > 1 and n_tot > 1.5 * self.n_iterations_last_prec_update)
                  or (p.physical_type == "incompressible stationary" and step % 6 == 1)):
                self.build_preconditioner()
                premature_update = True
            its, lin_res = self.solve_system(linear_tolerance)
            self.linear_iterations.append((its, lin_res))
            n_tot += its
            self.solution[0] += self.solution_update[0]
            self.solution[1] += self.solution_update[1]
            if not implicit:
                break                   # (the reference returns `step` = 0 here)
            res = self.compute_residual()
            if res < p.tol_nl_iteration:
                break
            step += 1
        # :941-970 does the linear solver deteriorate?
        no = ts.step_no()
        if self.update_preconditioner_frequency > 0 and no % (50 * self.update_preconditioner_frequency) == 0:
            self.update_preconditioner_frequency = 0
        if self.update_preconditioner:
            self.n_iterations_last_prec_update = n_tot
            self.time_step_last_prec_update = no
            self.update_preconditioner = False
        if n_tot > 1.2 * self.n_iterations_last_prec_update:
            if premature_update or n_tot > 2 * self.n_iterations_last_prec_update:
                self.update_preconditioner_frequency = no - self.time_step_last_prec_update
            self.update_preconditioner = True
        if (self.time_step_last_prec_update < 3 and no > 14) or no < 2:
            self.update_preconditioner = True
        if (not self.update_preconditioner and not premature_update and self.update_preconditioner_frequency > 0
                and no + 1 - self.time_step_last_prec_update >= self.update_preconditioner_frequency):
            self.update_preconditioner = True
        return step, n_tot

    def advance_time_step(self):
        self.init_time_advance()
        res = self.compute_residual()
        return self.solve_nonlinear_system(res)

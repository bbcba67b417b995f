"""CPU restatement of the two-phase time step of LevelSetOKZSolver (TEST INFRASTRUCTURE ONLY):
tests/rising_bubble.cc + tests/rising_bubble_ls.prm of the reference -- in 2D on its own 40 x 80
mesh (golden output), and the same algorithm in 3D as the checker of the device drivers.

    LevelSetBaseAlgorithm::advance_time_step        source/level_set_base.cc:190-291
    TwoPhaseBaseAlgorithm::init_time_advance        source/two_phase_base.cc:441-460
    advance_concentration / reinitialize / compute_normal / compute_curvature (+ correction) /
    compute_heaviside / compute_force               source/level_set_okz*.cc
    NavierStokes::compute_residual, Newton          source/navier_stokes.cc:781-960

All operators are the oracle's (oracle/adaflo_oracle.c); the Krylov recurrences are
oracle/krylov_oracle.py.  The Navier-Stokes Newton systems are solved EXACTLY: the Jacobian (the
oracle's vmult on the state its residual stored, with the variable density / viscosity arrays of
compute_force) is assembled by coloured probing and factorised by SciPy's sparse LU -- cheap in 2D.
A converged Newton iteration is independent of the linear solver, so the numbers the reference
prints at the START of the following steps (advection residual and iterations, reinitialisation
iterations, first Navier-Stokes residual) are reproducible; see tests/test_oracle_golden_ls.py."""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from . import krylov_oracle as ko
from . import oracle as orc


class RisingBubble:
    def __init__(self, time_stepping_factory, ncell=(40, 80), s=4, k=2, eps_rel=1.5, dt=0.02, no_slip_everywhere=False,
                 linearization=0, max_nl=10, domain=None, centre=None, radius=0.25, physics=None, n_initial_reinit=2):
        """dim = len(ncell); the domain is [0,1]^(dim-1) x [0,2] with gravity along the last axis"""
        self.s, self.k, self.eps_rel, self.dt = s, k, eps_rel, dt
        # linearization: 0 Newton, 1 Picard, 2 semi-implicit, 3 explicit convection (NSParams); max_nl = "NL max
        # iterations" (1 for the linear schemes of rising_bubble_ls_imex.prm / _expl.prm)
        self.linearization, self.max_nl = linearization, max_nl
        self.ncell = list(ncell)
        dim = self.dim = len(ncell)
        lower, upper = domain if domain is not None else ((0.,) * dim, (1.,) * (dim - 1) + (2.,))
        self.mesh = orc.Mesh.make(self.ncell, lower, upper)
        self.h = (upper[0] - lower[0]) / ncell[0]
        # tests/rising_bubble_ls.prm; other cases (tests/spurious_currents_ls.prm) pass their own
        self.physics = dict(surface_tension=0.0245, gravity=0.98, density=1.0, density_diff=-0.9, viscosity=0.01,
                            viscosity_diff=-0.009)
        self.physics.update(physics or {})
        centre = np.full(dim, 0.5) if centre is None else np.asarray(centre, dtype=float)
        self.eps_used = eps_rel / s * self.h                               # two_phase_base.cc:290-291
        mesh = self.mesh
        self.nn, self.nu, self.np_ = mesh.n_nodes(s), mesh.n_nodes(k) * dim, mesh.n_nodes(k - 1)
        self.nq_ls = (2 * s) ** dim
        # NavierStokes, advection and reinitialisation advance their own TimeStepping objects
        self.ts_ns, self.ts_adv, self.ts_rei = (time_stepping_factory() for _ in range(3))
        self.prm = self._ls_prm(self.ts_ns)
        # initialize_mass_matrix_diagonal: the curvature operator without diffusion is the mass matrix
        idx = np.indices(tuple(s * n + 1 for n in reversed(ncell)))      # (z,) y, x
        col = sum((idx[dim - 1 - d] % 2) * 2 ** d for d in range(dim)).reshape(-1)
        diag = np.zeros(self.nn)
        for c in range(2 ** dim):
            y = orc.ls_curvature_vmult(mesh, self.prm, (col == c).astype(float), apply_diffusion=False)
            diag[col == c] = y[col == c]
        self.inv_diag = 1.0 / diag
        # no-slip at the bottom / top, symmetry (normal component) on the side walls (rising_bubble.cc:133-150)
        if no_slip_everywhere:
            self.con_u = orc.boundary_mask(mesh, k, dim)
        else:
            self.con_u = orc.boundary_mask(mesh, k, dim, faces=[2 * (dim - 1), 2 * (dim - 1) + 1])
            for d in range(dim - 1):
                self.con_u = self.con_u | orc.boundary_mask(mesh, k, dim, faces=[2 * d, 2 * d + 1], comps=[d])
        x = orc.node_coordinates(mesh, s, fe_type=1)
        self.phi = -np.tanh((np.linalg.norm(x - centre, axis=1) - radius) / (2 * self.eps_used))   # rising_bubble.cc:59-77
        self.normal, self.kappa, self.kappa_old = np.zeros(dim * self.nn), np.zeros(self.nn), np.zeros(self.nn)
        self.normal_q = np.zeros(mesh.n_cells * self.nq_ls * dim)
        self.log = {}
        its = []
        if n_initial_reinit > 0:                                            # number initial reinit steps
            self.phi, its = self.reinitialize(self.phi, n_initial_reinit)
            self.ts_rei.next()
        self.log["initial_reinitialize"] = its
        self.phi_old, self.phi_oo = self.phi.copy(), self.phi.copy()
        self.u, self.p = np.zeros(self.nu), np.zeros(self.np_)
        self.u_old, self.u_oo, self.p_old = np.zeros(self.nu), np.zeros(self.nu), np.zeros(self.np_)

    def _ls_prm(self, ts):
        return orc.make_ls_params(self.s, self.eps_used, self.h, self.dt, ts.weight(), self.h, self.eps_rel)

    # ---- level-set steps
    def compute_normal(self, phi, fast):
        A = lambda v: orc.ls_normal_vmult(self.mesh, self.prm, v)
        rhs = orc.ls_normal_rhs(self.mesh, self.prm, phi)
        self.normal = ko.cg(A, rhs, x0=self.normal, inv_diag=np.tile(self.inv_diag, self.dim), max_it=4000,
                            rel_tol=1e-5 if fast else 1e-7)[0]

    def reinitialize(self, phi, steps):                                      # reinitialization.cc:255-375
        A = lambda v: orc.ls_reinit_vmult(self.mesh, self.prm, v, self.normal_q)
        its = []
        for tau in range(steps):
            if tau == 0:
                self.compute_normal(phi, True)
            rhs = orc.ls_reinit_rhs(self.mesh, self.prm, phi, self.normal, self.normal_q, diffuse_only=False,
                                    first_step=tau == 0)
            inc, it, *_ = ko.cg(A, rhs, inv_diag=self.inv_diag, max_it=2000, abs_tol=1e-50, rel_tol=1e-6)
            its.append(it)
            phi = phi + inc
        return phi, its

    def compute_force(self):                                                 # level_set_okz.cc:415-432
        mesh, prm = self.mesh, self.prm
        H = orc.ls_compute_heaviside(mesh, self.s, self.eps_rel, self.phi)
        self.compute_normal(self.phi, False)
        # compute_curvature.cc:350-355: the production solve does NOT use ComputeCurvatureMatrix (that call is
        # commented out) but the assembled projection matrix shared with the normal projection
        # (level_set_okz.cc:262-300: mass + 4 max(eps_used / eps, h / ls)^2 Laplace) = one scalar block of the
        # normal operator
        nn = self.nn

        def A(v):
            blocks = np.zeros(self.dim * nn)
            blocks[:nn] = v
            return orc.ls_normal_vmult(mesh, prm, blocks)[:nn].copy()
        kappa = ko.cg(A, orc.ls_curvature_rhs(mesh, prm, self.normal), x0=self.kappa, inv_diag=self.inv_diag,
                      max_it=2000, rel_tol=1e-8)[0]
        with np.errstate(divide="ignore", invalid="ignore"):                 # compute_curvature.cc:360-376
            dist = np.where(1 - self.phi ** 2 > 1e-2, self.eps_used * np.log((1 + self.phi) / (1 - self.phi)), 0.0)
        sel = kappa > 1e-4
        kappa[sel] = 1.0 / (1.0 / kappa[sel] + dist[sel] / (self.dim - 1))
        self.kappa = kappa
        return orc.ls_compute_force(mesh, self.s, self.k, H, kappa, **self.physics,
                                    interpolate_grad_onto_pressure=True, con_u=self.con_u)

    # ---- exact Newton step: Jacobian by coloured probing, sparse LU
    def _assemble(self, vm):
        mesh, k, nu, npp, dim = self.mesh, self.k, self.nu, self.np_, self.dim
        nnu, nnp = mesh.nodes_per_dim(k), mesh.nodes_per_dim(k - 1)
        iu, ip = np.indices(tuple(reversed(nnu))), np.indices(tuple(reversed(nnp)))
        Xu = [iu[dim - 1 - d].reshape(-1) for d in range(dim)]       # node coordinates (index units), x first
        Xp = [ip[dim - 1 - d].reshape(-1) for d in range(dim)]
        P, Pp = 3 * k, 3 * (k - 1)                          # probe columns 3 cells apart never share a row
        up, pu = (k - 1) / k, k / (k - 1)                   # velocity <-> pressure node index units
        colu = sum((Xu[d] % P) * P ** d for d in range(dim))
        colp = sum((Xp[d] % Pp) * Pp ** d for d in range(dim))
        near = lambda a, c, per: c + per * np.round((a - c) / per).astype(int)

        def flat(coords, nn):
            out = coords[dim - 1]
            for d in range(dim - 2, -1, -1):
                out = out * nn[d] + coords[d]
            return out
        rows, cols, vals = [], [], []
        for c in range(P ** dim):
            cc = [(c // P ** d) % P for d in range(dim)]
            sel = np.nonzero(colu == c)[0]
            if sel.size == 0:
                continue
            for comp in range(dim):
                e = np.zeros(nu)
                e[dim * sel + comp] = 1.0
                yu, yp = vm(e, np.zeros(npp))
                nz = np.nonzero(yu)[0]
                rows.append(nz)
                cols.append(dim * flat([near(Xu[d][nz // dim], cc[d], P) for d in range(dim)], nnu) + comp)
                vals.append(yu[nz])
                nz = np.nonzero(yp)[0]                      # pressure node I sits near velocity node I k / (k - 1)
                rows.append(nu + nz)
                cols.append(dim * flat([near(pu * Xp[d][nz], cc[d], P) for d in range(dim)], nnu) + comp)
                vals.append(yp[nz])
        for c in range(Pp ** dim):
            cc = [(c // Pp ** d) % Pp for d in range(dim)]
            if not np.any(colp == c):
                continue
            yu, yp = vm(np.zeros(nu), (colp == c).astype(float))
            nz = np.nonzero(yu)[0]
            rows.append(nz)
            cols.append(nu + flat([near(up * Xu[d][nz // dim], cc[d], Pp) for d in range(dim)], nnp))
            vals.append(yu[nz])
            nz = np.nonzero(yp)[0]
            rows.append(nu + nz)
            cols.append(nu + flat([near(Xp[d][nz], cc[d], Pp) for d in range(dim)], nnp))
            vals.append(yp[nz])
        n = nu + npp
        return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))

    def advance_time_step(self, tol_nl=1e-9):
        """returns what the reference prints for the step: advection (initial residual, iterations),
        reinitialisation iterations, Navier-Stokes residual history"""
        mesh, k, nu = self.mesh, self.k, self.nu
        ts = self.ts_ns
        ts.next()                                                           # init_time_advance
        u_new, p_new = ts.extrapolate(self.u, self.u_old), ts.extrapolate(self.p, self.p_old)
        self.u_oo, self.u_old, self.u = self.u_old, self.u, u_new.copy()
        self.p_old, self.p = self.p, p_new.copy()
        so, st = ts.old_step_size(), ts.step_size()
        new_phi, new_kappa = self.phi.copy(), self.kappa.copy()
        if so > 0:                                                          # two_phase_base.cc:449-452
            new_phi = (st + so) / so * self.phi - st / so * self.phi_old
            new_kappa = (st + so) / so * self.kappa - st / so * self.kappa_old
        self.phi_oo, self.phi_old, self.phi = self.phi_old, self.phi, new_phi
        self.kappa_old, self.kappa = self.kappa, new_kappa
        # advance_concentration (advance_concentration.cc:503-660)
        ta = self.ts_adv
        ta.next()
        self.prm = self._ls_prm(ta)
        uq = np.zeros(mesh.n_cells * self.nq_ls * self.dim)
        rhs = orc.ls_advect_rhs(mesh, self.prm, k, self.phi, self.phi_old, self.phi_oo, self.u, uq, ta.weight_old(),
                                ta.weight_old_old(), ta.scheme == "bdf_2" and ta.step_no() > 1)
        A = lambda v: orc.ls_advect_vmult(mesh, self.prm, v, uq)
        inc, adv_it, adv_r0, *_ = ko.bicgstab(A, rhs, inv_diag=self.inv_diag, max_it=30, abs_tol=0.05 * tol_nl, rel_tol=1e-8)
        self.phi = self.phi + inc
        # reinitialize (number reinit steps = 2)
        self.prm = self._ls_prm(self.ts_rei)
        self.phi, rei_its = self.reinitialize(self.phi, 2)
        self.ts_rei.next()
        # compute_force, Navier-Stokes
        self.prm = self._ls_prm(ts)
        force, rho, mu = self.compute_force()
        ph = self.physics
        nsp = orc.NSParams.make(linearization=self.linearization, beta=0.5, density=ph["density"], viscosity=ph["viscosity"],
                                density_diff=ph["density_diff"], weight=ts.weight(),
                                weight_old=ts.weight_old(), weight_old_old=ts.weight_old_old(), tau1=ts.tau1(),
                                extrap_old=ts.factor_extrapol_old, extrap_old_old=ts.factor_extrapol_old_old)
        lin, history = np.zeros(mesh.n_cells * (k + 1) ** self.dim * orc.n_lin(self.dim)), []
        damp = None if rho is None else np.zeros_like(rho)       # constant parameters: no coefficient arrays
        for it in range(self.max_nl + 1):
            ru, rp = orc.ns_residual(mesh, k, nsp, self.u, self.p, self.u_old, self.u_oo, con_u=self.con_u, lin=lin,
                                     rho=rho, mu=mu, damp=damp, user_u=force)
            rp = rp - rp.mean()          # mean-value projection of the pressure rows (uniform mesh: weights ~ 1 inside)
            history.append(float(np.hypot(np.linalg.norm(ru), np.linalg.norm(rp))))
            if history[-1] < tol_nl or it == self.max_nl:
                break
            if self.dim == 3:      # OpenMP sum-factorised restatement (checked against the naive one)
                vm = lambda a, b: tuple(v.copy() for v in orc.fast_ns_vmult(mesh, k, nsp, a, b, self.con_u, None, lin=lin,
                                                                            rho=rho, mu=mu, damp=damp))
            else:
                vm = lambda a, b: orc.ns_vmult(mesh, k, nsp, a, b, self.con_u, None, lin=lin, rho=rho, mu=mu, damp=damp)
            J = self._assemble(vm).tolil()
            J[nu, :] = 0
            J[nu, nu] = 1.0                                                   # pin one pressure value
            rhs_all = np.concatenate([ru, rp])
            rhs_all[nu] = 0.0
            d = spla.spsolve(J.tocsc(), rhs_all)
            self.u += d[:nu]
            self.p += d[nu:]
        return (adv_r0, adv_it), rei_its, history


def bubble_statistics_2d(sim):
    """TwoPhaseBaseAlgorithm<2>::compute_bubble_statistics (source/two_phase_base.cc:621-905): area, perimeter,
    circularity, mean velocity and centre of mass of the region phi > 0, with the interface located by linear
    interpolation on a (k+3) x (k+3) trapezoidal sub-grid of every cut cell -- the numbers the reference prints
    after every time step ("Degree of circularity", "Mean bubble velocity", "Position of the center of mass").
    Returns (circularity, mean velocity[2], centre[2], area)."""
    assert sim.dim == 2
    mesh, s, k = sim.mesh, sim.s, sim.k
    ncx, ncy = sim.ncell
    hx, hy = mesh.h[0], mesh.h[1]
    sub = k + 3
    pts = np.linspace(0.0, 1.0, sub + 1)                               # QIterated(QTrapezoid, k + 3)
    Sl, _ = orc.shape_1d(1, s, pts)                                     # [point][dof], FE_Q_iso_Q1(s)
    Sv, _ = orc.shape_1d(0, k, pts)
    xg, wg = orc.gauss_legendre(k)                                      # interior_quadrature = QGauss(k)
    Sg, _ = orc.shape_1d(0, k, xg)
    nls = [s * ncx + 1, s * ncy + 1]
    nvs = [k * ncx + 1, k * ncy + 1]
    phi = sim.phi.reshape(nls[1], nls[0])
    vel = sim.u.reshape(nvs[1], nvs[0], 2)
    area = perimeter = 0.0
    com, velocity = np.zeros(2), np.zeros(2)
    for cy in range(ncy):
        for cx in range(ncx):
            loc = phi[s * cy:s * cy + s + 1, s * cx:s * cx + s + 1]
            lv = vel[k * cy:k * cy + k + 1, k * cx:k * cx + k + 1]
            x0, y0 = mesh.origin[0] + hx * cx, mesh.origin[1] + hy * cy
            flat = loc.reshape(-1)
            if not np.any(flat[1:] * flat[0] <= 0):                     # :668-690 the interface does not cross the cell
                if flat[0] > 0:
                    ug = np.einsum("qj,pi,jic->qpc", Sg, Sg, lv)        # [qy][qx][c]
                    w = np.outer(wg, wg) * hx * hy
                    area += w.sum()
                    com += [np.sum(w * (x0 + hx * xg)[None, :]), np.sum(w * (y0 + hy * xg)[:, None])]
                    velocity += np.einsum("qp,qpc->c", w, ug)
                continue
            cval = Sl @ loc @ Sl.T                                      # [py][px]
            uval = np.einsum("qj,pi,jic->qpc", Sv, Sv, lv)
            wsub = hx * hy / (sub * sub) / 4.0                          # JxW * weight_correction of a patch corner
            for dy in range(sub):
                for dx in range(sub):
                    idx = [(dy, dx), (dy, dx + 1), (dy + 1, dx), (dy + 1, dx + 1)]
                    c = np.array([cval[i] for i in idx]) + 1e-22
                    quad = np.array([[x0 + hx * pts[i[1]], y0 + hy * pts[i[0]]] for i in idx])
                    local_area = 1.0
                    rx0 = rx1 = ry0 = ry1 = -1.0
                    px0 = px1 = py0 = py1 = None
                    if c[0] * c[1] <= 0:
                        rx0 = c[0] / (c[0] - c[1])
                        px0 = quad[0] + (quad[1] - quad[0]) * rx0
                    if c[2] * c[3] <= 0:
                        rx1 = c[2] / (c[2] - c[3])
                        px1 = quad[2] + (quad[3] - quad[2]) * rx1
                    if c[0] * c[2] <= 0:
                        ry0 = c[0] / (c[0] - c[2])
                        py0 = quad[0] + (quad[2] - quad[0]) * ry0
                    if c[1] * c[3] <= 0:
                        ry1 = c[1] / (c[1] - c[3])
                        py1 = quad[1] + (quad[3] - quad[1]) * ry1

                    def cut(my_area, corner, a, b):
                        nonlocal local_area, perimeter
                        local_area -= my_area if corner < 0 else 1 - my_area
                        perimeter += np.linalg.norm(a - b)
                    if rx0 > 0:
                        if ry0 > 0:
                            cut(0.5 * rx0 * ry0, c[0], px0, py0)
                        if ry1 > 0:
                            cut(0.5 * (1 - rx0) * ry1, c[1], px0, py1)
                        if rx1 > 0 and ry0 < 0 and ry1 < 0:
                            cut(0.5 * (rx0 + rx1), c[0], px0, px1)
                    if rx1 > 0:
                        if ry0 > 0:
                            cut(0.5 * rx1 * (1 - ry0), c[2], px1, py0)
                        if ry1 > 0:
                            cut(0.5 * (1 - rx1) * (1 - ry1), c[3], px1, py1)
                    if ry0 > 0 and ry1 > 0 and rx0 < 0 and rx1 < 0:
                        cut(0.5 * (ry0 + ry1), c[0], py0, py1)
                    if rx0 <= 0 and rx1 <= 0 and ry0 <= 0 and ry1 <= 0 and c[0] <= 0:
                        local_area = 0.0
                    my_area = local_area * wsub
                    for n, i in enumerate(idx):
                        area += my_area
                        com += quad[n] * my_area
                        velocity += uval[i] * my_area
    circularity = 2.0 * np.sqrt(area * np.pi) / perimeter
    return circularity, velocity / area, com / area, area


def spurious_current_statistics_2d(sim):
    """tests/spurious_currents.cc:121-222 -- what the reference prints after every step of the static-bubble test:
    the relative error of the pressure jump in per cent, ((mean p over the cells whose centre is closer than 0.1 to the
    origin) - (mean p over the boundary) - 2 (dim - 1) sigma) / (2 (dim - 1) sigma) * 100 with QGauss(k+1), and the
    largest velocity magnitude over the QIterated(QTrapezoid, k+2) points of all cells."""
    assert sim.dim == 2
    mesh, k = sim.mesh, sim.k
    ncx, ncy = sim.ncell
    hx, hy = mesh.h[0], mesh.h[1]
    sigma = sim.physics["surface_tension"]
    # largest velocity
    pts = np.linspace(0.0, 1.0, k + 3)
    Sv, _ = orc.shape_1d(0, k, pts)
    nvx, nvy = k * ncx + 1, k * ncy + 1
    vel = sim.u.reshape(nvy, nvx, 2)
    ix = (np.arange(ncx)[:, None] * k + np.arange(k + 1)[None, :])
    iy = (np.arange(ncy)[:, None] * k + np.arange(k + 1)[None, :])
    loc = vel[iy[:, :, None, None], ix[None, None, :, :]]                  # [cy][j][cx][i][c]
    uq = np.einsum("ajbic,qj,pi->aqbpc", loc, Sv, Sv)
    size = float(np.sqrt((uq ** 2).sum(axis=-1)).max())
    # pressure jump
    xg, wg = orc.gauss_legendre(k + 1)
    Sp, _ = orc.shape_1d(0, k - 1, xg)
    npx, npy = (k - 1) * ncx + 1, (k - 1) * ncy + 1
    pr = sim.p.reshape(npy, npx)
    kp = k - 1
    p_avg = one_avg = p_b = one_b = 0.0
    ends = {0: orc.shape_1d(0, kp, np.array([0.0]))[0][0], 1: orc.shape_1d(0, kp, np.array([1.0]))[0][0]}
    for cy in range(ncy):
        for cx in range(ncx):
            cxm, cym = mesh.origin[0] + hx * (cx + 0.5), mesh.origin[1] + hy * (cy + 0.5)
            lp = pr[kp * cy:kp * cy + kp + 1, kp * cx:kp * cx + kp + 1]        # [j][i]
            if np.hypot(cxm, cym) < 0.1:
                pq = Sp @ lp @ Sp.T
                w = np.outer(wg, wg) * hx * hy
                p_avg += float((pq * w).sum())
                one_avg += float(w.sum())
            for side, on_boundary, along, length in ((0, cx == 0, "y", hy), (1, cx == ncx - 1, "y", hy),
                                                     (0, cy == 0, "x", hx), (1, cy == ncy - 1, "x", hx)):
                if not on_boundary:
                    continue
                face = (lp @ ends[side]) if along == "y" else (ends[side] @ lp)    # dofs along the face
                pf = Sp @ face
                p_b += float((pf * wg).sum() * length)
                one_b += length
    jump = ((p_avg / one_avg - p_b / one_b) - 2.0 * (sim.dim - 1) * sigma) / (2.0 * (sim.dim - 1) * sigma) * 100.0
    return jump, size

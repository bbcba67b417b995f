"""CPU restatement of the nonlinear time step around the operators (TEST INFRASTRUCTURE ONLY).

NavierStokes::advance_time_step for the coupled implicit Newton scheme on a uniform brick:
  init_time_advance              source/navier_stokes.cc:659-745  (TimeStepping::next, shift of the
                                 old solutions, extrapolated initial guess, boundary values)
  compute_residual               :781-800  (residual, mean-value projection of the pressure rows)
  solve_nonlinear_system         :832-960  (Newton: J delta = -F, solution += delta)
  solve_system                   :561-653  (FGMRES(50) with the block preconditioner)
The reference preconditions with ILU / AMG (Trilinos) -- not reproducible here and irrelevant for
the converged solution.  This file uses the same block-triangular structure
(source/navier_stokes_preconditioner.cc: velocity block, then the pressure Schur complement
approximated by  S^-1 ~ M_p(1/(mu+tau))^-1 + L_p(1/(gamma rho))^-1 ) with Jacobi-preconditioned
CG / BiCGStab as the inner solvers.  A converged Newton iteration gives the same discrete solution
whatever the linear solver, which is what pins the oracle to the reference's SECOND time step
(tests/beltrami_3d.output:35)."""
import numpy as np

from . import krylov_oracle as ko
from . import oracle as orc


def fgmres(A, b, M, restart=50, max_it=60, tol=1e-10):
    """flexible GMRES (Saad 1993), right preconditioner M (may change from call to call);
    A, M: callables on flat vectors.  Returns x, iterations, residual norm."""
    x = np.zeros_like(b)
    r = b.copy()
    beta = np.linalg.norm(r)
    total = 0
    while total < max_it and beta > tol:
        m = min(restart, max_it - total)
        V = [r / beta]
        Z = []
        H = np.zeros((m + 1, m))
        g = np.zeros(m + 1)
        g[0] = beta
        cs, sn = np.zeros(m), np.zeros(m)
        kk = 0
        for j in range(m):
            Z.append(M(V[j]))
            w = A(Z[j])
            for i in range(j + 1):          # modified Gram-Schmidt
                H[i, j] = w @ V[i]
                w -= H[i, j] * V[i]
            H[j + 1, j] = np.linalg.norm(w)
            V.append(w / H[j + 1, j] if H[j + 1, j] > 0 else w)
            for i in range(j):              # previous Givens rotations
                t = cs[i] * H[i, j] + sn[i] * H[i + 1, j]
                H[i + 1, j] = -sn[i] * H[i, j] + cs[i] * H[i + 1, j]
                H[i, j] = t
            d = np.hypot(H[j, j], H[j + 1, j])
            cs[j], sn[j] = H[j, j] / d, H[j + 1, j] / d
            H[j, j], H[j + 1, j] = d, 0.0
            g[j + 1] = -sn[j] * g[j]
            g[j] = cs[j] * g[j]
            kk = j + 1
            total += 1
            if abs(g[j + 1]) <= tol:
                break
        y = np.linalg.solve(np.triu(H[:kk, :kk]), g[:kk])
        for i in range(kk):
            x += y[i] * Z[i]
        r = b - A(x)
        beta = np.linalg.norm(r)
    return x, total, beta


class BeltramiStepper:
    """state of tests/beltrami.cc on an n^3 mesh of [-1,1]^3, Q_k/Q_{k-1}, all-Dirichlet velocity"""

    def __init__(self, n, time_stepping, k=2, viscosity=1.0, beta=0.5):
        self.mesh = orc.Mesh.make([n] * 3, [-1.0] * 3, [1.0] * 3)
        self.k, self.ts, self.nu, self.beta = k, time_stepping, viscosity, beta
        self.xu, self.xp = orc.node_coordinates(self.mesh, k), orc.node_coordinates(self.mesh, k - 1)
        self.con_u = orc.boundary_mask(self.mesh, k, 3)
        self.w = orc.ns_pressure_mass_weight(self.mesh, k)
        self.nq = (k + 1) ** 3
        self.lin = np.zeros(self.mesh.n_cells * self.nq * 12)
        # tests/beltrami.cc:436-440: nodal interpolation of the exact solution at t = 0
        self.u, self.p = orc.beltrami_u(self.xu, 0.0, viscosity), orc.beltrami_p(self.xp, 0.0, viscosity)
        self.u_old, self.u_oldold = np.zeros_like(self.u), np.zeros_like(self.u)
        self.p_old = np.zeros_like(self.p)

    def params(self):
        ts = self.ts
        return orc.NSParams.make(beta=self.beta, viscosity=self.nu, weight=ts.weight(), weight_old=ts.weight_old(),
                                 weight_old_old=ts.weight_old_old(), tau1=ts.tau1(),
                                 extrap_old=ts.factor_extrapol_old, extrap_old_old=ts.factor_extrapol_old_old)

    def init_time_advance(self):
        ts = self.ts
        ts.next()
        # :672-686 both blocks: cur <- extrapolate(cur, old), old <- cur, old_old <- old
        u_new, p_new = ts.extrapolate(self.u, self.u_old), ts.extrapolate(self.p, self.p_old)
        self.u_oldold, self.u_old, self.p_old = self.u_old, self.u, self.p
        self.u, self.p = u_new.copy(), p_new.copy()
        # apply_boundary_conditions :1216-1257: Dirichlet values of the new time level
        ub = orc.beltrami_u(self.xu, ts.now(), self.nu)
        self.u[self.con_u == 1] = ub[self.con_u == 1]

    def residual(self):
        """-F(u) with the pressure rows projected, and the (u, grad u) state for the Jacobian"""
        prm = self.params()
        ru, rp = orc.ns_residual(self.mesh, self.k, prm, self.u, self.p, self.u_old, self.u_oldold,
                                 con_u=self.con_u, lin=self.lin)
        rp = orc.ns_pressure_projection(rp, self.w, np.ones_like(self.w))
        return ru, rp

    # ---- Jacobian and block preconditioner
    def jacobian(self):
        prm, nu_, np_ = self.params(), self.u.size, self.p.size
        ones = np.ones_like(self.w)
        buf = (np.empty(nu_), np.empty(np_))

        def A(x):
            du, dp = orc.fast_ns_vmult(self.mesh, self.k, prm, x[:nu_], x[nu_:], self.con_u, None, lin=self.lin,
                                       weights=self.w, modes=ones, out=buf)
            return np.concatenate([du, dp])
        return A, prm

    def velocity_diagonal(self, prm):
        """diagonal of the velocity block by probing: nodes 3 apart never share a Q2 cell"""
        nn = self.mesh.nodes_per_dim(self.k)
        idx = np.indices((nn[2], nn[1], nn[0]))
        colour = (idx[2] % 3) + 3 * (idx[1] % 3) + 9 * (idx[0] % 3)
        diag = np.zeros((nn[2], nn[1], nn[0], 3))
        zero_p = np.zeros_like(self.p)
        for c in range(27):
            sel = colour == c
            for d in range(3):
                e = np.zeros_like(diag)
                e[sel, d] = 1.0
                y, _ = orc.fast_ns_vmult(self.mesh, self.k, prm, e.reshape(-1), zero_p, self.con_u, None, lin=self.lin)
                diag[sel, d] = y.reshape(diag.shape)[sel, d]
        return diag.reshape(-1)

    def preconditioner(self, prm, inner_tol=1e-2):
        nu_ = self.u.size
        zero_p, zero_u = np.zeros_like(self.p), np.zeros_like(self.u)
        inv_du = 1.0 / self.velocity_diagonal(prm)
        mesh, k = self.mesh, self.k
        Mp = lambda v: orc.ns_pressure_mass_vmult(mesh, k, prm, v)
        Lp = lambda v: orc.ns_pressure_poisson_vmult(mesh, k, prm, v)
        inv_dm = 1.0 / Mp(np.ones_like(self.p))      # row sums: positive for Q1
        # diagonal of the Q1 Laplacian by probing with stride 2
        nn = mesh.nodes_per_dim(k - 1)
        idx = np.indices((nn[2], nn[1], nn[0]))
        col = (idx[2] % 2) + 2 * (idx[1] % 2) + 4 * (idx[0] % 2)
        dl = np.zeros(col.shape)
        for c in range(8):
            e = (col == c).astype(float)
            dl[col == c] = Lp(e.reshape(-1)).reshape(col.shape)[col == c]
        inv_dl = 1.0 / dl.reshape(-1)
        Av = lambda v: orc.fast_ns_vmult(mesh, k, prm, v, zero_p, self.con_u, None, lin=self.lin)[0].copy()

        def M(r):
            ru, rp = r[:nu_], r[nu_:]
            # lower block-triangular: A du = r_u, then -S dp = r_p - B du with B = (q, -div .)
            du, *_ = ko.bicgstab(Av, ru, inv_diag=inv_du, max_it=200, rel_tol=inner_tol)
            s = orc.ns_divergence_vmult_add(mesh, k, prm, du, zero_p.copy(), self.con_u, None) - rp
            s -= s.mean()                                                # consistent with the Neumann problem
            pm, *_ = ko.cg(Mp, s, inv_diag=inv_dm, max_it=50, rel_tol=inner_tol)
            pl, *_ = ko.cg(Lp, s, inv_diag=inv_dl, max_it=200, rel_tol=inner_tol)
            pl -= pl.mean()
            return np.concatenate([du, pm + pl])
        return M

    def advance_time_step(self, tol_nl=1e-9, max_nl=10, verbose=False):
        """returns the list of (||F_u||, ||F_p||) printed by the reference per Newton step"""
        self.init_time_advance()
        history = []
        nu_ = self.u.size
        M = None
        for step in range(max_nl + 1):
            ru, rp = self.residual()
            history.append((np.linalg.norm(ru), np.linalg.norm(rp)))
            res = np.hypot(*history[-1])
            if verbose:
                print("   %-11.3e %-12.3e" % history[-1])
            if res < tol_nl or step == max_nl:
                break
            A, prm = self.jacobian()
            if M is None:
                M = self.preconditioner(prm)
            delta, its, lin_res = fgmres(A, np.concatenate([ru, rp]), M, restart=50, max_it=80, tol=1e-7 * res)
            if verbose:
                print("      lin its %d res %.2e" % (its, lin_res))
            self.u += delta[:nu_]
            self.p += delta[nu_:]
        return history

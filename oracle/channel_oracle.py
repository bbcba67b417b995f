"""CPU restatement of the reference's single-phase channel tests in 2D (TEST INFRASTRUCTURE ONLY):
tests/poiseuille.cc (poiseuille_stokes.prm, poiseuille_ns.prm) and tests/couette.cc on their own meshes.

    NavierStokes::apply_boundary_conditions   source/navier_stokes.cc:1216-1310   Dirichlet values and the face integrals
                                              const_rhs_i = int_open (phi_i . n) p_ext dS, QGauss(k + 1) per face
    NavierStokes::compute_residual / Newton   source/navier_stokes.cc:781-960
    set_open_boundary_with_normal_flux        source/flow_base_algorithm.cc:140-155: tangential components constrained

Operators: oracle/adaflo_oracle.c; Newton systems solved exactly (Jacobian by coloured probing of the oracle's vmult,
SciPy sparse LU, as oracle/two_phase_oracle.py); no pressure mean projection (the open boundaries fix the level)."""
import types

import numpy as np
import scipy.sparse.linalg as spla

from . import oracle as orc
from . import two_phase_oracle as tpo


class ChannelFlow:
    def __init__(self, time_stepping, ncell=(64, 16), k=2, viscosity=0.5, stokes=False, p_ext=lambda x: 2.0 - x[:, 0],
                 wall_velocity=None):
        """[-2, 2] x [-1, 0]; faces 0 / 1 open with normal flux and pressure p_ext, face 2 no-slip; face 3: symmetry
        (wall_velocity None, tests/poiseuille.cc:248-255) or a Dirichlet wall moving with wall_velocity
        (tests/couette.cc:133-142)"""
        self.ts, self.k, self.viscosity, self.stokes, self.dim = time_stepping, k, viscosity, stokes, 2
        self.mesh = mesh = orc.Mesh.make(list(ncell), (-2.0, -1.0), (2.0, 0.0))
        self.nu, self.np_ = mesh.n_nodes(k) * 2, mesh.n_nodes(k - 1)
        con = orc.boundary_mask(mesh, k, 2, faces=[2]) | orc.boundary_mask(mesh, k, 2, faces=[0, 1], comps=[1])
        con |= orc.boundary_mask(mesh, k, 2, faces=[3], comps=[1] if wall_velocity is None else [0, 1])
        self.con_u = con
        self.x = orc.node_coordinates(mesh, k)
        self.dirichlet = np.zeros(self.nu)
        if wall_velocity is not None:
            top = np.abs(self.x[:, 1]) < 1e-13
            self.dirichlet.reshape(-1, 2)[top] = wall_velocity
        self.const_rhs = self._open_boundary_rhs(p_ext)
        self.const_rhs[con == 1] = 0.0                    # distribute_local_to_global skips constrained rows
        self.u, self.p = np.zeros(self.nu), np.zeros(self.np_)
        self.u_old, self.u_oo, self.p_old = np.zeros(self.nu), np.zeros(self.nu), np.zeros(self.np_)

    def _open_boundary_rhs(self, p_ext):
        mesh, k = self.mesh, self.k
        xg, wg = orc.gauss_legendre(k + 1)
        S, _ = orc.shape_1d(0, k, xg)                                     # [q][i]
        nnx, nny = mesh.nodes_per_dim(k)
        rhs = np.zeros((nny, nnx, 2))
        ncy, hy = mesh.ncell[1], mesh.h[1]
        for side, xf, normal in ((0, -2.0, -1.0), (1, 2.0, 1.0)):
            for cy in range(ncy):
                yq = mesh.origin[1] + hy * (cy + xg)
                pq = p_ext(np.stack([np.full_like(yq, xf), yq], axis=1))
                rhs[cy * k:cy * k + k + 1, -1 if side else 0, 0] += normal * hy * (S.T @ (wg * pq))
        return rhs.reshape(-1)

    def params(self):
        ts = self.ts
        return orc.NSParams.make(physical_type=2 if self.stokes else 0, beta=0.5, viscosity=self.viscosity,
                                 density=0.0 if self.stokes else 1.0, weight=ts.weight(), weight_old=ts.weight_old(),
                                 weight_old_old=ts.weight_old_old(), tau1=ts.tau1(), extrap_old=ts.factor_extrapol_old,
                                 extrap_old_old=ts.factor_extrapol_old_old)

    def advance_time_step(self, tol_nl=1e-11, max_nl=10):
        """returns the residual history of the step (what the reference prints in its `Nonlin Res` column)"""
        ts, mesh, k, nu = self.ts, self.mesh, self.k, self.nu
        ts.next()
        u_new, p_new = ts.extrapolate(self.u, self.u_old), ts.extrapolate(self.p, self.p_old)
        self.u_oo, self.u_old, self.u = self.u_old, self.u, u_new.copy()
        self.p_old, self.p = self.p, p_new.copy()
        self.u[self.con_u == 1] = self.dirichlet[self.con_u == 1]
        prm = self.params()
        lin, history = np.zeros(mesh.n_cells * (k + 1) ** 2 * 6), []
        helper = types.SimpleNamespace(mesh=mesh, k=k, nu=nu, np_=self.np_, dim=2)
        for it in range(max_nl + 1):
            ru, rp = orc.ns_residual(mesh, k, prm, self.u, self.p, self.u_old, self.u_oo, con_u=self.con_u, lin=lin,
                                     rhs_u=self.const_rhs)
            history.append(float(np.hypot(np.linalg.norm(ru), np.linalg.norm(rp))))
            if history[-1] < tol_nl or it == max_nl:
                break
            vm = lambda a, b: orc.ns_vmult(mesh, k, prm, a, b, self.con_u, None, lin=lin)
            J = tpo.RisingBubble._assemble(helper, vm)
            d = spla.spsolve(J.tocsc(), np.concatenate([ru, rp]))
            self.u += d[:nu]
            self.p += d[nu:]
        return history


class Flow1D:
    """tests/1d_flow.cc (1d_flow.prm, 1d_flow_damped.prm): NavierStokes<1> on [0, 2.5] with 2048 cells, velocity 2 at
    t = 0, open ends with the pressures 2 and 1 (in 1D the face integrals are the point values -2 and +1 on the first and
    the last velocity row), tau grad div = 1e-5, optional damping; exact Newton steps as in ChannelFlow"""

    def __init__(self, time_stepping, n=2048, k=2, viscosity=0.01, damping=0.0, tau_grad_div=1e-5):
        self.ts, self.k, self.viscosity, self.damping, self.tau_grad_div = time_stepping, k, viscosity, damping, tau_grad_div
        self.mesh = orc.Mesh.make([n], [0.0], [2.5])
        self.nu, self.np_ = self.mesh.n_nodes(k), self.mesh.n_nodes(k - 1)
        self.u, self.p = np.full(self.nu, 2.0), np.zeros(self.np_)
        self.u_old, self.u_oo, self.p_old = np.zeros(self.nu), np.zeros(self.nu), np.zeros(self.np_)
        self.const_rhs = np.zeros(self.nu)
        self.const_rhs[0], self.const_rhs[-1] = -2.0, 1.0

    def advance_time_step(self, tol_nl=1e-12, max_nl=8):
        ts, mesh, k, nu = self.ts, self.mesh, self.k, self.nu
        ts.next()
        u_new, p_new = ts.extrapolate(self.u, self.u_old), ts.extrapolate(self.p, self.p_old)
        self.u_oo, self.u_old, self.u = self.u_old, self.u, u_new.copy()
        self.p_old, self.p = self.p, p_new.copy()
        # (the oracle's NSParams carries the damping with the stored, flipped sign: parameters.cc:466-467)
        prm = orc.NSParams.make(beta=0.5, viscosity=self.viscosity, damping=-self.damping, tau_grad_div=self.tau_grad_div,
                                weight=ts.weight(), weight_old=ts.weight_old(), weight_old_old=ts.weight_old_old(),
                                tau1=ts.tau1(), extrap_old=ts.factor_extrapol_old, extrap_old_old=ts.factor_extrapol_old_old)
        lin, history = np.zeros(mesh.n_cells * (k + 1) * 2), []
        helper = types.SimpleNamespace(mesh=mesh, k=k, nu=nu, np_=self.np_, dim=1)
        for it in range(max_nl + 1):
            ru, rp = orc.ns_residual(mesh, k, prm, self.u, self.p, self.u_old, self.u_oo, lin=lin, rhs_u=self.const_rhs)
            history.append(float(np.hypot(np.linalg.norm(ru), np.linalg.norm(rp))))
            if history[-1] < tol_nl or it == max_nl:
                break
            J = tpo.RisingBubble._assemble(helper, lambda a, b: orc.ns_vmult(mesh, k, prm, a, b, None, None, lin=lin))
            d = spla.spsolve(J.tocsc(), np.concatenate([ru, rp]))
            self.u += d[:nu]
            self.p += d[nu:]
        return history

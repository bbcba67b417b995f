"""Shared set-up for the parity tests: the same seeded inputs for the oracle
(oracle/) and for the HIP engine behind the C ABI."""
import numpy as np

import adaflo_amd
from oracle import oracle as orc

PHYS = {0: "incompressible", 1: "incompressible stationary", 2: "stokes"}
LIN = {0: "coupled implicit Newton", 1: "coupled implicit Picard",
       2: "coupled velocity semi-implicit", 3: "coupled velocity explicit", 4: "projection"}
BETA = {1.0: "conservative", 0.0: "convective", 0.5: "skew-symmetric"}


def rel_l2(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


class Case:
    """One operator configuration, realised for both sides."""

    def __init__(self, ncell, k=2, lower=(-1., -1., -1.), upper=(1., 1., 1.), faces_u=range(6),
                 faces_p=(), physical_type=0, linearization=0, beta=0.5, tau_grad_div=0.0,
                 density=1.0, viscosity=1.0, damping=0.0, density_diff=0.0, dt=0.05,
                 steps=2, pressure_average_fix=True, seed=20260515):
        self.ncell, self.k = list(ncell), k
        self.faces_u, self.faces_p = list(faces_u), list(faces_p)
        self.mesh = orc.Mesh.make(self.ncell, lower, upper)
        self.lower, self.upper = lower, upper
        self.rng = np.random.default_rng(seed)
        # parameters on the engine side go through the reference-shaped classes
        self.fp = adaflo_amd.FlowParameters(
            velocity_degree=k, physical_type=PHYS[physical_type], linearization=LIN[linearization],
            formulation_convective_term=BETA[beta], viscosity=viscosity, density=density,
            damping=damping, tau_grad_div=tau_grad_div, density_diff=density_diff,
            time_step_size_start=dt, end_time=100 * dt)
        self.ts = adaflo_amd.TimeStepping(self.fp)
        for _ in range(steps):
            self.ts.next()
        ts = self.ts
        self.prm = orc.NSParams.make(
            physical_type=physical_type, linearization=linearization, beta=beta,
            tau_grad_div=tau_grad_div, density=self.fp.density, viscosity=viscosity,
            damping=-damping, density_diff=density_diff, weight=ts.weight(),
            weight_old=ts.weight_old(), weight_old_old=ts.weight_old_old(), tau1=ts.tau1(),
            extrap_old=ts.factor_extrapol_old, extrap_old_old=ts.factor_extrapol_old_old)
        self.pressure_average_fix = pressure_average_fix
        self.con_u = orc.boundary_mask(self.mesh, k, 3, faces=self.faces_u)
        self.con_p = orc.boundary_mask(self.mesh, k - 1, 1, faces=self.faces_p)
        self.n_u = self.mesh.n_nodes(k) * 3
        self.n_p = self.mesh.n_nodes(k - 1)
        self.nq = (k + 1) ** 3
        self.n_cells = self.mesh.n_cells

    # ---- inputs
    def random_u(self):
        return self.rng.uniform(-1, 1, self.n_u)

    def random_p(self):
        return self.rng.uniform(-1, 1, self.n_p)

    def smooth_u(self, t=0.0):
        return orc.beltrami_u(orc.node_coordinates(self.mesh, self.k), t)

    def smooth_p(self, t=0.0):
        return orc.beltrami_p(orc.node_coordinates(self.mesh, self.k - 1), t)

    def random_lin(self):
        return self.rng.uniform(-1, 1, self.n_cells * self.nq * 12)

    def random_coefficients(self):
        n = self.n_cells * self.nq
        return (self.rng.uniform(0.5, 2.0, n), self.rng.uniform(0.5, 2.0, n),
                self.rng.uniform(-0.5, 0.5, n))

    def weights_modes(self):
        if not self.pressure_average_fix:
            return None, None
        w = orc.ns_pressure_mass_weight(self.mesh, self.k, self.con_p)
        modes = np.ones(self.n_p)
        modes[self.con_p == 1] = 0.0
        return w, modes

    # ---- engine
    def engine(self, device=0):
        mesh = adaflo_amd.BrickMesh(self.ncell, self.lower, self.upper)
        op = adaflo_amd.NavierStokesMatrix(self.fp, mesh, dirichlet_faces_u=self.faces_u,
                                           constrained_faces_p=self.faces_p, device=device)
        op.initialize(self.ts, self.pressure_average_fix)
        return op


def l2_norm_of_difference(mesh, k, dofs, ncomp, exact, n_gauss):
    """VectorTools::integrate_difference(..., QGauss(n_gauss), L2_norm) for a nodal FE_Q(k) field on the
    brick: sqrt(sum_cells sum_q |u_h(x_q) - exact(x_q)|^2 JxW)"""
    xg, wg = orc.gauss_legendre(n_gauss)
    S, _ = orc.shape_1d(0, k, xg)                      # [q][i]
    nn = mesh.nodes_per_dim(k)
    u = dofs.reshape(nn[2], nn[1], nn[0], ncomp)
    n = [mesh.ncell[d] for d in range(3)]
    h = [mesh.h[d] for d in range(3)]
    # values at all Gauss points, cell by cell along each axis: [cz][qz][cy][qy][cx][qx][c]
    idx = [np.arange(n[d])[:, None] * k + np.arange(k + 1)[None, :] for d in range(3)]
    loc = u[idx[2][:, :, None, None, None, None], idx[1][None, None, :, :, None, None], idx[0][None, None, None, None, :, :]]
    val = np.einsum("azbycxm,qz,ry,sx->aqbrcsm", loc, S, S, S)
    axes = [mesh.origin[d] + h[d] * (np.arange(n[d])[:, None] + xg[None, :]) for d in range(3)]   # [cell][q]
    Z, Y, X = np.meshgrid(axes[2].reshape(-1), axes[1].reshape(-1), axes[0].reshape(-1), indexing="ij")
    ex = exact(np.stack([X.reshape(-1), Y.reshape(-1), Z.reshape(-1)], axis=1)).reshape(val.shape)
    w = np.einsum("q,r,s->qrs", wg, wg, wg) * h[0] * h[1] * h[2]
    return np.sqrt(np.einsum("aqbrcsm,qrs->", (val - ex) ** 2, w))

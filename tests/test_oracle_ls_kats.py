"""Analytic known-answer tests for the level-set part of the oracle:
no kernel-level golden exists in the reference, so the restatement is pinned by exactness
properties of the weak forms (source/level_set_okz_*.cc)."""
import numpy as np
import pytest

from oracle import oracle as orc


def setup(s=2, ncell=(3, 2, 4), upper=(1.0, 0.5, 2.0)):
    mesh = orc.Mesh.make(list(ncell), [0.0] * 3, list(upper))
    h = [mesh.h[d] for d in range(3)]
    prm = orc.make_ls_params(s, 1.5 * max(h) / s, min(h), 0.02, 75.0, max(h), 1.5)
    x = orc.node_coordinates(mesh, s, fe_type=1)
    return mesh, prm, x, float(np.prod(upper))


@pytest.mark.parametrize("s", [1, 2, 4])
def test_advection_of_a_linear_profile_is_exact(s):
    mesh, prm, x, vol = setup(s)
    a = np.array([0.3, -1.2, 0.7])
    phi = x @ a + 0.25
    u = np.array([0.5, 0.25, -2.0])
    nq = (2 * s) ** 3
    uq = np.tile(u, mesh.n_cells * nq)
    dst = orc.ls_advect_vmult(mesh, prm, phi, uq)
    # hats sum to one: sum_i (w_i, gamma phi + u.grad phi) = integral over the box
    mean_phi = a @ (0.5 * np.array([1.0, 0.5, 2.0])) + 0.25
    assert abs(dst.sum() - (prm.weight * mean_phi + u @ a) * vol) < 1e-11 * abs(dst.sum())


def test_normal_rhs_and_mass_rows():
    mesh, prm, x, vol = setup(2)
    a = np.array([0.3, -1.2, 0.7])
    rhs = orc.ls_normal_rhs(mesh, prm, x @ a).reshape(3, -1)
    assert np.allclose(rhs.sum(axis=1), a * vol, rtol=1e-12)
    ones = np.ones(3 * mesh.n_nodes(2))
    d = orc.ls_normal_vmult(mesh, prm, ones).reshape(3, -1)
    assert np.allclose(d.sum(axis=1), vol, rtol=1e-12)       # (w,1) + (grad w, delta grad 1)
    c = orc.ls_curvature_vmult(mesh, prm, np.ones(mesh.n_nodes(2)), apply_diffusion=True)
    assert abs(c.sum() - vol) < 1e-12 * vol


def test_reinitialization_operator_with_constant_normal_is_an_anisotropic_laplacian():
    s = 2
    mesh, prm, x, vol = setup(s)
    nq = (2 * s) ** 3
    n = np.array([0.0, 0.0, 1.0])
    nrm = np.tile(n, mesh.n_cells * nq)
    # phi depending on x only: n.grad phi = 0 -> only the mass part (w, phi/dtau) remains
    phi = 0.5 * x[:, 0] + 0.1
    d = orc.ls_reinit_vmult(mesh, prm, phi, nrm, diffuse_only=False)
    dtau_inv = max(0.95 / (1.0 / 9.0 * prm.minimal_edge_length / s), 1.0 / (5.0 * prm.time_step))
    assert abs(d.sum() - dtau_inv * (0.5 * 0.5 + 0.1) * vol) < 1e-11 * abs(d.sum())
    # symmetric operator
    a, b = np.random.default_rng(0).uniform(-1, 1, (2, phi.size))
    da = orc.ls_reinit_vmult(mesh, prm, a, nrm, diffuse_only=False)
    db = orc.ls_reinit_vmult(mesh, prm, b, nrm, diffuse_only=False)
    assert abs(a @ db - b @ da) < 1e-12 * abs(a @ db)


def test_tanh_profile_is_a_fixed_point_of_the_reinitialization_rhs():
    """phi = tanh(x / (2 eps)), n = e_x: 1/2 (1 - phi^2) - eps n.grad phi = 0
    (level_set_okz_reinitialization.cc:176-178); the discrete rhs vanishes at O(h^2)."""
    norms = []
    for n in (4, 8):
        s = 2
        mesh = orc.Mesh.make([n, 2, 2], [-1.0, 0, 0], [1.0, 0.5, 0.5])
        eps = 0.4                                   # resolved profile, larger than h/s
        prm = orc.make_ls_params(s, eps, 2.0 / n, 0.02, 1.0, 2.0 / n, 1.5)
        x = orc.node_coordinates(mesh, s, fe_type=1)
        phi = np.tanh(x[:, 0] / (2 * eps))
        normal = np.zeros((3, phi.size))
        normal[0] = 1.0
        nq = np.zeros(mesh.n_cells * (2 * s) ** 3 * 3)
        rhs = orc.ls_reinit_rhs(mesh, prm, phi, normal.reshape(-1), nq, diffuse_only=False, first_step=True)
        interior = orc.boundary_mask(mesh, s, 1, faces=[0, 1]) == 0
        norms.append(np.abs(rhs[interior]).max())
        # a profile of the wrong width is NOT a fixed point: its rhs is an order of magnitude larger
        bad = orc.ls_reinit_rhs(mesh, prm, np.tanh(x[:, 0] / (4 * eps)), normal.reshape(-1), nq,
                                diffuse_only=False, first_step=True)
        assert np.abs(bad[interior]).max() > 10 * norms[-1]
    assert norms[1] < norms[0] / 4.0      # pointwise defect O(h) times |grad w| h^3 / h


def test_curvature_rhs_of_a_radial_normal_field():
    """n = x - x0 (unnormalised): div(n/|n|) = 2/r; check the integral of the rhs over a shell-free box."""
    s = 2
    mesh = orc.Mesh.make([6, 6, 6], [1.0, 1.0, 1.0], [2.0, 2.0, 2.0])
    prm = orc.make_ls_params(s, 0.1, 1 / 6, 0.02, 1.0, 1 / 6, 1.5)
    x = orc.node_coordinates(mesh, s, fe_type=1)
    normal = np.ascontiguousarray(x.T).reshape(-1)           # blocks: n_x, n_y, n_z with x0 = 0
    rhs = orc.ls_curvature_rhs(mesh, prm, normal)
    # -int div(x/|x|) = -int 2/|x| over [1,2]^3, by tensor Gauss quadrature
    g, w = np.polynomial.legendre.leggauss(12)
    g, w = 1.5 + 0.5 * g, 0.5 * w
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    W = w[:, None, None] * w[None, :, None] * w[None, None, :]
    exact = -(2.0 / np.sqrt(X * X + Y * Y + Z * Z) * W).sum()
    assert abs(rhs.sum() - exact) < 2e-3 * abs(exact)       # piecewise-linear normalised normal


def test_convection_stabilization_is_a_consistent_diffusion_term():
    """known answers of the stabilisation terms (level_set_okz_advance_concentration.cc:248-249 + :419-472):
    with a constant artificial viscosity the cell term (grad w, nu grad v) minus the boundary term
    (w, n . nu grad v) is -(w, nu laplace v): zero for a linear field on every row, and equal to the weak
    Laplacian tested with w on interior rows for a quadratic field; the maximal velocity of a constant field
    is its norm; the artificial viscosity follows :361-365"""
    mesh = orc.Mesh.make([3, 2, 2], [0., 0., 0.], [1., 1., 2.])
    s = 2
    h = [mesh.h[d] for d in range(3)]
    prm = orc.make_ls_params(s, 0.1, min(h), 0.01, 0.0, max(h), 1.5)          # weight 0: no mass term
    x = orc.node_coordinates(mesh, s, 1)
    nq = (2 * s) ** 3
    uq0 = np.zeros(mesh.n_cells * nq * 3)
    nu = np.full(mesh.n_cells, 0.7)
    lin = 1 + 2 * x[:, 0] - x[:, 1] + 0.5 * x[:, 2]
    assert np.abs(orc.ls_advect_vmult(mesh, prm, lin, uq0, art_visc=nu)).max() < 1e-13
    # symmetry faces are left out of the boundary term: the rows on face 0 (n = -e_x, area 2) then keep the
    # cell term alone, (w, n . nu grad v) = -0.7 * 2 per unit area
    out = orc.ls_advect_vmult(mesh, prm, lin, uq0, art_visc=nu, symmetry=1)
    on_face = x[:, 0] == 0
    assert np.abs(out[~on_face]).max() < 1e-13 and abs(out[on_face].sum() + 0.7 * 2 * 2.0) < 1e-12
    vel = np.tile([0.3, -0.4, 1.2], mesh.n_nodes(2))
    assert abs(orc.ls_max_velocity(mesh, 2, vel) - 1.3) < 1e-14
    # artificial viscosity: u_old + u_old_old = 2 u, phi linear and steady: residual = |2 u . 2 grad phi| / 4
    uq, nu_out = np.zeros(mesh.n_cells * nq * 3), np.zeros(mesh.n_cells)
    orc.ls_advect_rhs(mesh, prm, 2, lin, lin, lin, vel, uq, -1.0, 0.0, True, vel_old=vel, vel_old_old=vel,
                      old_step_size=0.01, global_scaling=10.0, art_visc=nu_out)
    resid = abs(0.3 * 2 - 0.4 * (-1) + 1.2 * 0.5)
    assert np.allclose(nu_out, 0.03 * 2.6 * max(h) * min(1.0, resid / 10.0), rtol=1e-13)

"""CPU tier: known answers for the oracle's compute_heaviside / compute_force restatement
(source/level_set_okz.cc:317-413, :479-540; include/adaflo/level_set_base.h:122-158)."""
import numpy as np
import pytest

from oracle import oracle as orc


def test_discrete_heaviside_is_the_integral_of_peskins_delta():
    x = np.linspace(-2.5, 2.5, 2001)
    h = orc.discrete_heaviside(x)
    assert h[0] == 0.0 and h[-1] == 1.0 and abs(h[1000] - 0.5) < 1e-15
    assert np.all(np.diff(h) >= -1e-15)                       # monotone
    # d/dx = discrete_delta (level_set_base.h:148-158)
    xm = 0.5 * (x[1:] + x[:-1])
    a = np.abs(xm)
    delta = np.where(a > 2, 0.0, np.where(a > 1, (5 - 2 * a - np.sqrt(np.maximum(-7 + 12 * a - 4 * a * a, 0))) / 8,
                                          (3 - 2 * a + np.sqrt(np.maximum(1 + 4 * a - 4 * a * a, 0))) / 8))
    assert np.abs(np.diff(h) / np.diff(x) - delta).max() < 2e-3
    assert abs(float(np.sum(0.5 * (delta[1:] + delta[:-1]) * np.diff(xm))) - 1.0) < 1e-5


@pytest.mark.parametrize("s", [2, 4])
def test_heaviside_of_a_tanh_profile(s):
    mesh = orc.Mesh.make([4, 4, 4], (0., 0., 0.), (1., 1., 1.))
    x = orc.node_coordinates(mesh, s, fe_type=1).reshape(-1, 3)
    eps_rel, h = 1.5, 0.25
    eps_used = eps_rel * h / s
    d = x[:, 0] - 0.5
    phi = np.tanh(d / (2 * eps_used))
    H = orc.ls_compute_heaviside(mesh, s, eps_rel, phi)
    assert H.min() == 0.0 and H.max() == 1.0
    assert np.allclose(H[np.abs(d) < 1e-14], 0.5)
    # inside the band: H = discrete_heaviside(log((1+phi)/(1-phi)) * 2 eps / s) = dh(d / eps_used * 2 eps / s)
    band = np.abs(phi) < np.tanh(2) * 0.999
    assert np.allclose(H[band], orc.discrete_heaviside(d[band] / eps_used * eps_rel * 2 / s), atol=1e-13)


@pytest.mark.parametrize("on_pressure", [False, True])
def test_force_of_a_linear_heaviside_and_constant_curvature(on_pressure):
    s, k = 2, 2
    mesh = orc.Mesh.make([3, 2, 4], (0., 0., 0.), (1.5, 1., 2.))
    x = orc.node_coordinates(mesh, s, fe_type=1).reshape(-1, 3)
    Hl = 0.2 + 0.3 * x[:, 0] - 0.1 * x[:, 2]                  # in the iso-Q1 space: exact gradient
    kappa = np.full(x.shape[0], 2.0)
    sigma, g, rho0, drho = 0.7, 9.81, 1.0, 0.5
    f, rho, mu = orc.ls_compute_force(mesh, s, k, Hl, kappa, surface_tension=sigma, gravity=g, density=rho0,
                                      density_diff=drho, viscosity=0.1, viscosity_diff=0.2,
                                      interpolate_grad_onto_pressure=on_pressure)
    f = f.reshape(-1, 3)
    vol = 1.5 * 1.0 * 2.0
    # partition of unity of the velocity basis: sum_i (phi_i, F) = int F
    assert abs(f[:, 0].sum() - sigma * 2.0 * 0.3 * vol) < 1e-12
    assert abs(f[:, 1].sum()) < 1e-12
    mean_H = 0.2 + 0.3 * 0.75 - 0.1 * 1.0
    assert abs(f[:, 2].sum() - (sigma * 2.0 * (-0.1) - g * (rho0 + drho * mean_H)) * vol) < 1e-11
    # rho, mu at the quadrature points are the affine images of H there
    assert abs(rho.mean() - (rho0 + drho * mean_H)) < 1e-12 and abs(mu.mean() - (0.1 + 0.2 * mean_H)) < 1e-12
    assert rho.min() >= rho0 + drho * Hl.min() - 1e-12 and rho.max() <= rho0 + drho * Hl.max() + 1e-12

"""Analytic known-answer tests of the oracle and consistency of its two
implementations (naive tables vs sum-factorised OpenMP)."""
import numpy as np
import pytest

from common import Case, rel_l2
from oracle import oracle as orc


def test_1d_data_against_numpy():
    for n in range(1, 7):
        x, w = orc.gauss_legendre(n)
        xr, wr = np.polynomial.legendre.leggauss(n)
        assert np.allclose(x, 0.5 * (xr + 1), atol=1e-15) and np.allclose(w, 0.5 * wr, atol=1e-15)
    assert np.allclose(orc.gauss_lobatto(3), [0, 0.5, 1])
    assert np.allclose(orc.gauss_lobatto(4), [0, 0.5 - 0.5 / np.sqrt(5), 0.5 + 0.5 / np.sqrt(5), 1])
    assert np.allclose(orc.gauss_lobatto(5), [0, 0.5 - 0.5 * np.sqrt(3 / 7), 0.5, 0.5 + 0.5 * np.sqrt(3 / 7), 1])
    for k in (1, 2, 3, 4):
        xq, _ = orc.gauss_legendre(k + 1)
        S, D = orc.shape_1d(0, k, xq)
        assert np.allclose(S.sum(axis=1), 1.0) and np.allclose(D.sum(axis=1), 0.0, atol=1e-13)
        assert np.allclose(S @ orc.gauss_lobatto(k + 1), xq)  # linear reproduction


@pytest.mark.parametrize("dim", [2, 3])
def test_poiseuille_is_in_the_kernel_of_the_stokes_operator(dim):
    """u = (1/(2 nu))(1-y^2) e_x, p = 2 - x (tests/poiseuille.cc:73-113) is in the Q2/Q1 space:
    interior rows of the Stokes residual vanish to round-off."""
    nu = 0.1
    ncell = [4, 6] if dim == 2 else [3, 4, 2]
    mesh = orc.Mesh.make(ncell, [-1.0] * dim, [1.0] * dim)
    xu, xp = orc.node_coordinates(mesh, 2), orc.node_coordinates(mesh, 1)
    u = np.zeros((xu.shape[0], dim))
    u[:, 0] = 0.5 / nu * (1 - xu[:, 1] ** 2)
    p = 2 - xp[:, 0]
    prm = orc.NSParams.make(physical_type=2, density=0.0, viscosity=nu)
    con_u = orc.boundary_mask(mesh, 2, dim)
    du, dp = orc.ns_vmult(mesh, 2, prm, u.reshape(-1).copy(), p.copy(), None, None)
    assert np.abs(du[con_u == 0]).max() < 1e-12      # momentum rows of interior test functions
    assert np.abs(dp).max() < 1e-12                  # continuity rows: div u = 0 exactly


def test_constant_pressure_and_rigid_translation():
    case = Case((3, 2, 4), k=2, upper=(1.0, 0.8, 1.7), pressure_average_fix=False, faces_u=[])
    ones_p = np.ones(case.n_p)
    du, dp = orc.ns_vmult(case.mesh, 2, case.prm, np.zeros(case.n_u), ones_p, None, None,
                          lin=np.zeros(case.n_cells * 27 * 12))
    interior = orc.boundary_mask(case.mesh, 2, 3) == 0
    assert np.abs(du[interior]).max() < 1e-13        # B^T 1 = 0 on interior rows
    # rigid translation: viscous part 0, mass part gamma * rho * M 1 -> total = gamma * rho * volume
    u = np.tile([1.0, 0.0, 0.0], case.n_u // 3)
    du, dp = orc.ns_vmult(case.mesh, 2, case.prm, u, np.zeros(case.n_p), None, None,
                          lin=np.zeros(case.n_cells * 27 * 12))
    vol = 2.0 * 1.8 * 2.7
    assert abs(du.reshape(-1, 3)[:, 0].sum() - case.prm.weight * vol) < 1e-10
    assert np.abs(dp).max() < 1e-12


def test_stokes_operator_is_symmetric():
    case = Case((3, 3, 2), k=2, physical_type=2, pressure_average_fix=False)
    x = (case.random_u(), case.random_p())
    y = (case.random_u(), case.random_p())
    for v in (x, y):  # symmetric on the space with homogeneous constraints
        v[0][case.con_u == 1] = 0
    ax = orc.ns_vmult(case.mesh, 2, case.prm, *x, case.con_u, None)
    ay = orc.ns_vmult(case.mesh, 2, case.prm, *y, case.con_u, None)
    lhs = x[0] @ ay[0] + x[1] @ ay[1]
    rhs = y[0] @ ax[0] + y[1] @ ax[1]
    assert abs(lhs - rhs) < 1e-11 * abs(lhs)


def test_operator_is_linear_and_matches_unit_vector_probing():
    case = Case((2, 2, 2), k=2, pressure_average_fix=False)
    lin = case.random_lin()
    x, y = (case.random_u(), case.random_p()), (case.random_u(), case.random_p())
    f = lambda v: np.concatenate(orc.ns_vmult(case.mesh, 2, case.prm, v[0], v[1], case.con_u, None, lin=lin))
    z = (2.5 * x[0] - y[0], 2.5 * x[1] - y[1])
    assert rel_l2(f(z), 2.5 * f(x) - f(y)) < 1e-13


@pytest.mark.parametrize("k,ncell", [(2, (5, 4, 3)), (3, (3, 2, 2)), (4, (2, 2, 2))])
@pytest.mark.parametrize("lin", [0, 1, 3])
def test_fast_oracle_matches_naive_oracle(k, ncell, lin):
    case = Case(ncell, k=k, linearization=lin, faces_p=[2])
    su, sp, l = case.random_u(), case.random_p(), case.random_lin()
    rho, mu, dmp = case.random_coefficients()
    w, modes = case.weights_modes()
    kw = dict(lin=l, rho=rho, mu=mu, damp=dmp, weights=w, modes=modes)
    a = orc.ns_vmult(case.mesh, k, case.prm, su, sp, case.con_u, case.con_p, **kw)
    b = orc.fast_ns_vmult(case.mesh, k, case.prm, su, sp, case.con_u, case.con_p, **kw)
    assert rel_l2(b[0], a[0]) < 1e-13 and rel_l2(b[1], a[1]) < 1e-13


@pytest.mark.parametrize("k,ncell", [(2, (5, 4, 3)), (4, (2, 2, 2))])
@pytest.mark.parametrize("lin,phys", [(0, 0), (1, 0), (0, 1)])
def test_fast_oracle_residual_matches_naive_oracle(k, ncell, lin, phys):
    """orc_fast_ns_residual (OpenMP, sum-factorised; the full-size checker of the path bench.py times) against the naive
    restatement of NavierStokesMatrix::residual (navier_stokes_matrix.cc:266-293): right-hand side with a user vector, the
    quadrature-point state it writes, variable coefficients"""
    case = Case(ncell, k=k, linearization=lin, physical_type=phys, faces_u=[0, 2, 3, 5], faces_p=[1], tau_grad_div=0.2,
                damping=0.1, density=1.2, steps=3)
    su, sp, ou, oou = case.random_u(), case.random_p(), case.random_u(), case.random_u()
    uu, up = case.random_u(), case.random_p()
    rho, mu, dmp = case.random_coefficients()
    la, lb = np.zeros(case.n_cells * case.nq * 12), np.zeros(case.n_cells * case.nq * 12)
    kw = dict(con_u=case.con_u, con_p=case.con_p, rho=rho, mu=mu, damp=dmp, user_u=uu, user_p=up)
    a = orc.ns_residual(case.mesh, k, case.prm, su, sp, ou, oou, lin=la, **kw)
    b = orc.fast_ns_residual(case.mesh, k, case.prm, su, sp, ou, oou, lin=lb, **kw)
    assert rel_l2(b[0], a[0]) < 1e-13 and rel_l2(b[1], a[1]) < 1e-13
    assert np.abs(la).max() > 0.1 and rel_l2(lb, la) < 1e-13


def test_beltrami_residual_converges_with_mesh_refinement():
    """manufactured solution (tests/beltrami.cc:82-172): the discrete momentum residual of the
    exact fields (incl. time derivative via BDF weights of the exact history) decays under refinement"""
    errs = []
    for n in (4, 8):
        mesh = orc.Mesh.make([n] * 3, [-1.0] * 3, [1.0] * 3)
        xu, xp = orc.node_coordinates(mesh, 2), orc.node_coordinates(mesh, 1)
        dt, t = 1e-3, 0.1
        prm = orc.NSParams.make(weight=1.5 / dt, weight_old=-2 / dt, weight_old_old=0.5 / dt, beta=0.0)
        ru, rp = orc.ns_residual(mesh, 2, prm, orc.beltrami_u(xu, t), orc.beltrami_p(xp, t),
                                 orc.beltrami_u(xu, t - dt), orc.beltrami_u(xu, t - 2 * dt),
                                 con_u=orc.boundary_mask(mesh, 2, 3),
                                 lin=np.zeros(mesh.n_cells * 27 * 12))
        # dual norm proxy: residual functional scaled by the lumped mass (h^3)
        errs.append(np.linalg.norm(ru) / (2.0 / n) ** 1.5)
    assert errs[1] < errs[0] / 3.0


def test_config1_2d_beltrami_64x64_vmult_is_the_jacobian_of_the_residual():
    """BASELINE configs[0]: 2D Q2/Q1 on a uniform 64 x 64 mesh (CPU plumbing case, SURVEY 8 table: 4 096 cells,
    33 282 + 4 225 DoF).  The Newton vmult on the state stored by the residual is the derivative of the
    residual: (r(u - eps d) - r(u + eps d)) / (2 eps) = J d to O(eps^2) (the residual returns -F)."""
    mesh = orc.Mesh.make([64, 64], [-1.0, -1.0], [1.0, 1.0])
    k, dt = 2, 0.01
    assert mesh.n_cells == 4096 and 2 * mesh.n_nodes(k) == 33282 and mesh.n_nodes(k - 1) == 4225
    xu, xp = orc.node_coordinates(mesh, k), orc.node_coordinates(mesh, k - 1)
    u, p = orc.beltrami_u(xu, 0.0), orc.beltrami_p(xp, 0.0)
    u_old = orc.beltrami_u(xu, -dt)
    con_u = orc.boundary_mask(mesh, k, 2)
    prm = orc.NSParams.make(beta=0.5, weight=1.5 / dt, weight_old=-2.0 / dt, weight_old_old=0.5 / dt)
    nlin = mesh.n_cells * (k + 1) ** 2 * orc.n_lin(2)
    lin = np.zeros(nlin)
    orc.ns_residual(mesh, k, prm, u, p, u_old, u_old, con_u=con_u, lin=lin)
    rng = np.random.default_rng(7)
    du, dp = rng.uniform(-1, 1, u.size), rng.uniform(-1, 1, p.size)
    du[con_u == 1] = 0.0
    ju, jp = orc.ns_vmult(mesh, k, prm, du, dp, con_u, None, lin=lin)
    eps = 1e-5
    scratch = np.zeros(nlin)
    rpu, rpp = orc.ns_residual(mesh, k, prm, u + eps * du, p + eps * dp, u_old, u_old, con_u=con_u, lin=scratch)
    rmu, rmp = orc.ns_residual(mesh, k, prm, u - eps * du, p - eps * dp, u_old, u_old, con_u=con_u, lin=scratch)
    fd_u, fd_p = (rmu - rpu) / (2 * eps), (rmp - rpp) / (2 * eps)
    free = con_u == 0
    assert np.linalg.norm(fd_u[free] - ju[free]) < 1e-7 * np.linalg.norm(ju[free])
    assert np.linalg.norm(fd_p - jp) < 1e-7 * np.linalg.norm(jp)
    # linearity of the operator at this size
    a, b = 0.7, -1.3
    du2, dp2 = rng.uniform(-1, 1, u.size), rng.uniform(-1, 1, p.size)
    ju2, jp2 = orc.ns_vmult(mesh, k, prm, du2, dp2, con_u, None, lin=lin)
    ju3, jp3 = orc.ns_vmult(mesh, k, prm, a * du + b * du2, a * dp + b * dp2, con_u, None, lin=lin)
    assert np.linalg.norm(ju3 - (a * ju + b * ju2)) < 1e-12 * np.linalg.norm(ju3)
    assert np.linalg.norm(jp3 - (a * jp + b * jp2)) < 1e-12 * np.linalg.norm(jp3)


@pytest.mark.parametrize("k,ncell,phys,lin", [(2, (5, 4, 3), 0, 0), (2, (4, 4, 4), 0, 1), (3, (3, 2, 3), 0, 0), (4, (2, 3, 2), 0, 2),
                                              (2, (3, 3, 3), 2, 0), (5, (2, 2, 1), 0, 0), (2, (9, 1, 1), 0, 3), (4, (3, 3, 3), 0, 0)])
def test_cell_batched_cpu_baseline_equals_the_naive_oracle(k, ncell, phys, lin):
    """oracle/adaflo_oracle_batched.c (bench.py's cpu_baseline: W cells per SIMD register, batched state, even-odd
    kernels) against the naive oracle: ragged last batches, constrained velocity and pressure rows, mean projection"""
    mesh = orc.Mesh.make(list(ncell), [-1.0] * 3, [1.0, 2.0, 1.5])
    prm = orc.NSParams.make(physical_type=phys, linearization=lin, weight=30.0, tau_grad_div=0.3, viscosity=0.2,
                            density=1.1, damping=0.1)
    rng = np.random.default_rng(k)
    nu, npr = mesh.n_nodes(k) * 3, mesh.n_nodes(k - 1)
    su, sp = rng.uniform(-1, 1, nu), rng.uniform(-1, 1, npr)
    lin_q = rng.uniform(-1, 1, mesh.n_cells * (k + 1) ** 3 * 12)
    con_u = orc.boundary_mask(mesh, k, 3, faces=[0, 1, 2, 5])
    con_p = orc.boundary_mask(mesh, k - 1, 1, faces=[3])
    w, modes = orc.ns_pressure_mass_weight(mesh, k, con_p), np.ones(npr)
    ref_u, ref_p = orc.ns_vmult(mesh, k, prm, su, sp, con_u, con_p, lin=lin_q, weights=w, modes=modes)
    batched = orc.BatchedNSVmult(mesh, k, con_u, con_p, lin_q)
    assert batched.width in (2, 4, 8)
    for even_odd in (False, True):
        got_u, got_p = batched.vmult(prm, su, sp, weights=w, modes=modes, even_odd=even_odd)
        assert np.linalg.norm(got_u - ref_u) < 1e-13 * np.linalg.norm(ref_u)
        assert np.linalg.norm(got_p - ref_p) < 1e-13 * np.linalg.norm(ref_p)

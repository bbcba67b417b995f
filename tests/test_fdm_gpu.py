"""Fast-diagonalisation inverses behind the inner solves of the block preconditioner (csrc/fdm.hip, SURVEY 8f
rank 3): exact inverses of the constant-coefficient pressure mass / Poisson operators (checked against the engine's
own operators) and of c_m M + c_l K on the velocity space (checked against a Kronecker-product assembly in numpy),
and their effect on the Beltrami time step: the velocity-block BiCGStab needs a handful of iterations instead of
dozens while the reference's output lines are still reproduced (tests/test_navier_stokes_gpu.py)."""
import ctypes as C

import numpy as np
import pytest

import adaflo_amd
from adaflo_amd import _lib, beltrami
from adaflo_amd.navier_stokes import NavierStokes, gauss_lobatto_points, node_coordinates
from common import Case, rel_l2

pytestmark = pytest.mark.gpu


def _fdm(op, field, src, c_mass, c_lap):
    ctx = op._require()
    n = src.size
    s, d = adaflo_amd.DeviceVector.from_numpy(ctx, src), adaflo_amd.DeviceVector(ctx, n)
    _lib.check(ctx, _lib.load().adaflo_fdm_apply(ctx, field, d.ptr, s.ptr, c_mass, c_lap))
    return d.numpy()


@pytest.mark.parametrize("k,ncell,faces_p", [(2, (5, 4, 6), ()), (2, (9, 3, 4), (0,)), (3, (3, 4, 2), (1, 4)), (2, (1, 1, 1), ()),
                                             (2, (1, 3, 2), (0, 1)), (2, (2, 1, 3), (2, 3, 5)),
                                             # >= 192 nodes in a direction: the folded (even / odd) transforms of a symmetric 1D
                                             # problem -- natural ends, Dirichlet ends (padding modes), one Dirichlet end (not
                                             # symmetric: plain transform), an even number of nodes
                                             (2, (200, 2, 3), ()), (2, (2, 193, 2), (2, 3)), (2, (3, 2, 256), (4,)), (2, (191, 2, 2), (0, 1)),
                                             (3, (2, 3, 100), ())])
def test_pressure_mass_and_poisson_are_inverted_exactly(k, ncell, faces_p):
    case = Case(ncell, k=k, faces_p=faces_p, upper=(1.0, 0.7, 1.5), viscosity=0.3, tau_grad_div=0.2)
    op = case.engine()
    rng = np.random.default_rng(4)
    x = rng.uniform(-1, 1, case.n_p)
    xs, ys = op.initialize_p_vector(x), op.initialize_p_vector()
    op.pressure_mass_vmult(ys, xs)                                   # 1 / (mu + tau_gd) * M, constrained rows identity
    assert rel_l2(_fdm(op, 1, ys.numpy(), 1.0 / (0.3 + 0.2), 0.0), x) < 1e-10
    op.pressure_poisson_vmult(ys, xs)                                # 1 / (gamma rho) * K
    c = 1.0 / (case.ts.weight() * 1.0)
    got = _fdm(op, 1, ys.numpy(), 0.0, c)
    if faces_p:
        assert rel_l2(got, x) < 1e-9
    else:                                                             # pure Neumann: pseudo-inverse, constant dropped
        zs = op.initialize_p_vector(got)
        op.pressure_poisson_vmult(xs, zs)
        assert rel_l2(xs.numpy(), ys.numpy()) < 1e-9


def _mass_stiffness_1d(k, n, h):
    """FE_Q(k) on n cells with QGauss(k+1), as csrc/fdm.hip assembles them (independent numpy restatement)"""
    xg, wg = np.polynomial.legendre.leggauss(k + 1)
    xg, wg = 0.5 * (xg + 1), 0.5 * wg
    nodes = gauss_lobatto_points(k + 1)
    S = np.array([[np.prod([(x - nodes[j]) / (nodes[i] - nodes[j]) for j in range(k + 1) if j != i]) for i in range(k + 1)] for x in xg])
    D = np.array([[sum(np.prod([(x - nodes[j]) / (nodes[i] - nodes[j]) for j in range(k + 1) if j not in (i, m)]) / (nodes[i] - nodes[m])
                       for m in range(k + 1) if m != i) for i in range(k + 1)] for x in xg])
    N = k * n + 1
    M, K = np.zeros((N, N)), np.zeros((N, N))
    for c in range(n):
        sl = slice(c * k, c * k + k + 1)
        M[sl, sl] += h * np.einsum("q,qi,qj->ij", wg, S, S)
        K[sl, sl] += np.einsum("q,qi,qj->ij", wg, D, D) / h
    return M, K


@pytest.mark.parametrize("k,ncell,faces_u", [(2, (3, 2, 4), range(6)), (2, (4, 3, 2), (0, 3)), (3, (2, 2, 3), (4, 5)),
                                             (2, (96, 1, 2), range(6)), (2, (1, 100, 1), (0, 1)), (2, (1, 2, 97), (4,))])
def test_velocity_space_inverse_against_kronecker_assembly(k, ncell, faces_u):
    case = Case(ncell, k=k, faces_u=faces_u, upper=(1.0, 0.7, 1.5))
    op = case.engine()
    cm, cl = 3.5, 0.8
    mats = [_mass_stiffness_1d(k, ncell[d], case.mesh.h[d]) for d in range(3)]
    (Mx, Kx), (My, Ky), (Mz, Kz) = mats
    rng = np.random.default_rng(5)
    nn = [k * n + 1 for n in ncell]
    x = rng.uniform(-1, 1, (nn[2], nn[1], nn[0], 3))

    def apply(v):                                                   # (cm M + cl K) per component, [z][y][x]
        e = lambda A, B, Cc: np.einsum("zk,yj,xi,kji->zyx", A, B, Cc, v)
        return cm * e(Mz, My, Mx) + cl * (e(Mz, My, Kx) + e(Mz, Ky, Mx) + e(Kz, My, Mx))
    con = case.con_u.reshape(nn[2], nn[1], nn[0], 3).astype(bool)
    xin = np.where(con, 0.0, x)                                     # constrained columns are decoupled
    y = np.stack([apply(xin[..., c]) for c in range(3)], axis=-1)
    y = np.where(con, x, y)                                         # constrained rows: identity
    got = _fdm(op, 0, y.reshape(-1), cm, cl).reshape(x.shape)
    assert rel_l2(got, x) < 1e-10


def test_beltrami_time_step_with_fast_diagonalisation():
    """three ways through NavierStokes::solve_system on the 16^3 Beltrami case, all reproducing the reference's
    output line of time step #2: (1, 50) fast-diagonalisation inverses with the reference's two-stage strategy --
    the cheap solver without inner solves converges before `lin its before inner solvers` = 50 --, (1, 0) inner
    solves from the start (<= 10 BiCGStab iterations per velocity solve), (1, 4) the switch to the
    solver with inner solves after four cheap iterations, (0, .) Jacobi inner solves as in round 1"""
    nu = 1.0
    mesh = adaflo_amd.BrickMesh([16] * 3, [-1.0] * 3, [1.0] * 3)
    xu, xp = node_coordinates(mesh, 2), node_coordinates(mesh, 1)
    stats = {}
    for inner, before in ((1, 50), (1, 0), (1, 4), (0, 50)):
        fp = adaflo_amd.FlowParameters(velocity_degree=2, viscosity=nu, time_step_size_start=0.05, end_time=1.0,
                                       max_nl_iteration=10, tol_nl_iteration=1e-9, max_lin_iteration=100, tol_lin_iteration=1e-5,
                                       iterations_before_inner_solvers=before)
        ns = NavierStokes(fp, mesh, adaflo_amd.TimeStepping(fp), dirichlet_function=lambda x, t: beltrami.velocity(x, t, nu))
        ctx = ns.navier_stokes_matrix._require()
        _lib.check(ctx, _lib.load().adaflo_ns_preconditioner_set_inner(ctx, inner))
        ns.cheap_velocity_iterations = 0            # (0, .): Jacobi inner solves to their tolerance, no cheap stage
        ns.set_initial_condition(beltrami.velocity(xu, 0.0, nu).reshape(-1), beltrami.pressure(xp, 0.0, nu))
        ns.advance_time_step()
        assert np.hypot(*ns.history[-1]) < 1e-9
        ns.history.clear()
        ns.advance_time_step()
        assert "%.3e" % ns.history[0][0] == "2.348e+00" and "%.3e" % ns.history[0][1] == "5.678e-02"   # beltrami_3d.output:35
        solves, its = C.c_int64(), C.c_int64()
        _lib.check(ctx, _lib.load().adaflo_ns_preconditioner_statistics(ctx, C.byref(solves), C.byref(its)))
        stats[(inner, before)] = (its.value / max(solves.value, 1), [i for i, _ in ns.linear_iterations])
    cheap, strong, switch, jacobi = stats[(1, 50)], stats[(1, 0)], stats[(1, 4)], stats[(0, 50)]
    assert cheap[0] == 0.0 and max(cheap[1]) < 50, stats                        # never needs the inner solves
    assert 0.0 < strong[0] <= 10.0 and strong[0] < 0.5 * jacobi[0], stats
    assert sum(strong[1]) <= sum(jacobi[1]) + 5, stats
    assert switch[0] > 0.0 and all(4 < i < max(cheap[1]) + 1 for i in switch[1] if i > 4), stats


@pytest.mark.parametrize("s,ncell,upper", [(1, (5, 4, 3), (1.0, 1.0, 1.0)), (2, (4, 3, 5), (1.0, 0.7, 1.5)), (3, (2, 3, 2), (0.9, 1.2, 1.0)),
                                           (4, (3, 2, 2), (1.0, 0.5, 0.8)), (1, (1, 1, 1), (1.0, 1.0, 1.0)), (2, (1, 5, 1), (0.3, 1.0, 0.2)),
                                           (4, (48, 1, 2), (1.0, 0.1, 0.2)), (3, (1, 2, 65), (0.1, 0.2, 1.0))])
def test_projection_matrix_of_the_level_set_space_is_inverted_exactly(s, ncell, upper):
    """field 2 (FE_Q_iso_Q1(s), analytic cosine modes): adaflo_ls_projection_solve inverts the projection matrix of
    the normal / curvature solves (one scalar block of compute_normal_vmult = adaflo_ls_projection_vmult), and
    adaflo_fdm_apply(2, ...) inverts the curvature operator with its own damping"""
    from adaflo_amd import level_set_okz as lso
    mesh = adaflo_amd.BrickMesh(list(ncell), [0.0, 0.0, 0.0], list(upper))
    ops = lso.LevelSetOperators(mesh, s)
    eps_used, epsilon = 1.5 * max(mesh.h) / s, 1.5
    ops.set_parameters(eps_used, 0.02, 75.0, -100.0, 25.0, epsilon)
    lib, ctx = _lib.load(), ops._ctx
    rng = np.random.default_rng(11)
    x3 = rng.uniform(-1, 1, 3 * ops.n_dofs)
    src, dst, back = ops.vector(x3, blocks=3), ops.vector(blocks=3), ops.vector(blocks=3)
    lso.LevelSetOKZSolverComputeNormal(ops).compute_normal_vmult(dst, src)
    _lib.check(ctx, lib.adaflo_ls_projection_solve(ctx, back.ptr, dst.ptr, 3))
    assert rel_l2(back.numpy(), x3) < 1e-10
    x = x3[:ops.n_dofs]
    src1, dst1, back1 = ops.vector(x), ops.vector(), ops.vector()
    lso.LevelSetOKZSolverComputeCurvature(ops).compute_curvature_vmult(dst1, src1, True)
    b = max(eps_used / epsilon, max(mesh.h) / s)
    _lib.check(ctx, lib.adaflo_fdm_apply(ctx, 2, back1.ptr, dst1.ptr, 1.0, b * b))
    assert rel_l2(back1.numpy(), x) < 1e-10


def test_projection_solve_refuses_a_constrained_level_set_space():
    """constrained rows of the projection operator carry the user's diagonal, not the identity: the exact solve is
    only offered for the unconstrained space (the drivers then keep the CG path)"""
    from adaflo_amd import level_set_okz as lso
    mesh = adaflo_amd.BrickMesh([3, 3, 3], [0.0] * 3, [1.0] * 3)
    ops = lso.LevelSetOperators(mesh, 2, constrained_faces=(0,))
    ops.set_parameters(1.5 * max(mesh.h) / 2, 0.02, 75.0, -100.0, 25.0, 1.5)
    v, w = ops.vector(np.ones(ops.n_dofs)), ops.vector()
    code = _lib.load().adaflo_ls_projection_solve(ops._ctx, w.ptr, v.ptr, 1)
    assert code != 0 and b"unconstrained" in _lib.load().adaflo_last_error(ops._ctx)


@pytest.mark.parametrize("s,ncell", [(4, (16, 16, 32)), (2, (32, 64, 32)), (1, (64, 128, 256)), (4, (16, 128, 16)), (4, (256, 16, 16)),
                                     # 5 2^m intervals per direction: the reference's meshes (5 x 10 coarse cells refined)
                                     (4, (20, 20, 40)), (4, (40, 20, 20)), (2, (40, 80, 160)), (4, (20, 160, 20)), (4, (20, 16, 40)),
                                     # 3 2^m intervals
                                     (4, (24, 24, 48)), (2, (48, 96, 192)), (4, (96, 24, 20)), (4, (24, 192, 16))])
def test_fast_cosine_transforms_of_the_level_set_space(s, ncell, monkeypatch):
    """2^m, 3 2^m or 5 2^m intervals per direction and natural ends: the transforms run as fast cosine transforms in LDS
    (csrc/fdm_dct_kernel.hpp; 65 ... 1025 nodes per line, all thirteen lengths over the cases).  The result is the inverse
    of the projection matrix, and equal to what the matrix products give (ADAFLO_FDM_NO_DCT) to rounding"""
    from adaflo_amd import level_set_okz as lso
    mesh = adaflo_amd.BrickMesh(list(ncell), [0.0, 0.0, 0.0], [1.0, 0.7, 1.5])
    ops = lso.LevelSetOperators(mesh, s)
    eps_used, epsilon = 1.5 * max(mesh.h) / s, 1.5
    ops.set_parameters(eps_used, 0.02, 75.0, -100.0, 25.0, epsilon)
    lib, ctx = _lib.load(), ops._ctx
    x = np.random.default_rng(5).uniform(-1, 1, ops.n_dofs)
    src, dst, back, back_mm = ops.vector(x), ops.vector(), ops.vector(), ops.vector()
    lso.LevelSetOKZSolverComputeCurvature(ops).compute_curvature_vmult(dst, src, True)
    b = max(eps_used / epsilon, max(mesh.h) / s)
    _lib.check(ctx, lib.adaflo_fdm_apply(ctx, 2, back.ptr, dst.ptr, 1.0, b * b))
    assert rel_l2(back.numpy(), x) < 1e-10
    monkeypatch.setenv("ADAFLO_FDM_NO_DCT", "1")
    _lib.check(ctx, lib.adaflo_fdm_apply(ctx, 2, back_mm.ptr, dst.ptr, 1.0, b * b))
    monkeypatch.delenv("ADAFLO_FDM_NO_DCT")
    assert rel_l2(back_mm.numpy(), x) < 1e-10
    assert rel_l2(back.numpy(), back_mm.numpy()) < 1e-11
    _lib.check(ctx, lib.adaflo_fdm_apply(ctx, 2, dst.ptr, dst.ptr, 1.0, b * b))          # in place
    assert np.array_equal(dst.numpy(), back.numpy())


def test_fast_cosine_transforms_of_the_q1_pressure_space():
    """the 65 x 65 x 129 pressure grid of the two-phase benchmark: mass matrix and pure-Neumann Laplacian (pseudo-inverse)"""
    case = Case((64, 64, 128), k=2, upper=(1.0, 1.0, 2.0), viscosity=0.3, tau_grad_div=0.2)
    op = case.engine()
    x = np.random.default_rng(4).uniform(-1, 1, case.n_p)
    xs, ys = op.initialize_p_vector(x), op.initialize_p_vector()
    op.pressure_mass_vmult(ys, xs)
    assert rel_l2(_fdm(op, 1, ys.numpy(), 1.0 / (0.3 + 0.2), 0.0), x) < 1e-10
    op.pressure_poisson_vmult(ys, xs)
    got = _fdm(op, 1, ys.numpy(), 0.0, 1.0 / case.ts.weight())
    zs = op.initialize_p_vector(got)
    op.pressure_poisson_vmult(xs, zs)
    assert rel_l2(xs.numpy(), ys.numpy()) < 1e-9


@pytest.mark.parametrize("ncell,env", [((64, 64, 128), None), ((20, 20, 40), None), ((5, 4, 6), None), ((64, 32, 16), "1")])
def test_sum_of_two_inverses_in_one_application(ncell, env, monkeypatch):
    """adaflo_fdm_apply_sum: pressure mass + pressure Poisson (pseudo-)inverse of the Schur complement approximation as ONE
    fast-diagonalisation application (what the block preconditioner runs for constant coefficients) = the sum of the two
    separate applications, through the cosine transforms and through the matrix products (small grid / ADAFLO_FDM_NO_DCT)"""
    if env:
        monkeypatch.setenv("ADAFLO_FDM_NO_DCT", env)
    case = Case(ncell, k=2, upper=(1.0, 1.0, 2.0), viscosity=0.3, tau_grad_div=0.2)
    op = case.engine()
    ctx = op._require()
    x = np.random.default_rng(8).uniform(-1, 1, case.n_p)
    c_pm, c_pl = 1.0 / (0.3 + 0.2), 1.0 / case.ts.weight()
    ref = _fdm(op, 1, x, c_pm, 0.0) + _fdm(op, 1, x, 0.0, c_pl)
    s, d = adaflo_amd.DeviceVector.from_numpy(ctx, x), adaflo_amd.DeviceVector(ctx, x.size)
    _lib.check(ctx, _lib.load().adaflo_fdm_apply_sum(ctx, 1, d.ptr, s.ptr, c_pm, 0.0, 0.0, c_pl))
    assert rel_l2(d.numpy(), ref) < 1e-12


@pytest.mark.parametrize("k,ncell", [(3, (6, 5, 4)), (4, (4, 3, 5)), (5, (3, 3, 2))])
def test_sum_of_two_inverses_with_a_singular_part_of_higher_pressure_degree(k, ncell):
    """pressure degree >= 2: the null eigenvalue of the pure-Neumann Poisson part comes out of the generalised
    eigenvalue problem as round-off (not an exact zero as for Q1), so it has to be compared with the scale of ITS OWN
    operator -- with c_pm ~ c_pl (stationary problems) a threshold shared with the mass part let the constant mode through,
    multiplied by ~1e10.  The fused application must equal the two separate ones and stay mean-free in the
    Poisson part."""
    case = Case(ncell, k=k, viscosity=0.5, tau_grad_div=0.5)
    op = case.engine()
    ctx = op._require()
    x = np.random.default_rng(11).uniform(-1, 1, case.n_p) + 3.0          # a large constant component
    for c_pm, c_pl in [(1.0, 1.0), (1e-3, 1.0), (1.0, 1e3)]:
        ref_m, ref_l = _fdm(op, 1, x, c_pm, 0.0), _fdm(op, 1, x, 0.0, c_pl)
        s, d = adaflo_amd.DeviceVector.from_numpy(ctx, x), adaflo_amd.DeviceVector(ctx, x.size)
        _lib.check(ctx, _lib.load().adaflo_fdm_apply_sum(ctx, 1, d.ptr, s.ptr, c_pm, 0.0, 0.0, c_pl))
        got = d.numpy()
        assert np.all(np.isfinite(got))
        assert rel_l2(got, ref_m + ref_l) < 1e-11
        # the pseudo-inverse part has no component along the constant mode: K (got - M^-1 part) reproduces K K^+ x, and
        # its size is that of the data, not 1e10 times it
        assert np.linalg.norm(got - ref_m) < 1e3 * np.linalg.norm(x) / min(c_pl, 1.0)

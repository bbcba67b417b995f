"""GPU tier, dim = 2 (and dim = 1 at the end): the reference's own test suite is two-dimensional (tests/rising_bubble*.prm, spurious_currents*.prm),
so this is where the HIP kernels meet the reference's printed numbers directly.  The engine runs the generic
kernels with a FLAT third direction (one node, one quadrature point of weight 1: csrc/fe_kernels.hpp, SumFac<.., ZF>);
vectors keep three velocity components per node (the third one constrained), quadrature-point arrays keep their
3-vector / 12-double records with zeros in the slots of the missing direction.

    - operators against the committed 2D fixture and against the oracle's 2D operators (1e-12)
    - the two-phase drivers against tests/golden/reference_outputs.json = the lines of
      tests/rising_bubble_ls{,_picard,_imex,_expl,_q3}.output and tests/spurious_currents_ls.output"""
import json
import os
import types

import numpy as np
import pytest

import adaflo_amd
from adaflo_amd import level_set_okz as lso
from adaflo_amd.level_set_okz_solver import LevelSetOKZSolver
from common import BETA, LIN, PHYS, rel_l2
from golden_util import FixedTimeStepping, load, prm_dict
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-12
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_outputs.json")


def _ref(case):
    with open(GOLDEN) as f:
        return json.load(f)[case]


# ---- 2-component (oracle, reference) <-> 3-component (device) layouts -----------------------------------------------
def u3(u2):
    a = np.zeros((u2.size // 2, 3))
    a[:, :2] = u2.reshape(-1, 2)
    return a.reshape(-1)


def u2(u3_):
    a = np.asarray(u3_).reshape(-1, 3)
    assert np.all(a[:, 2] == 0.0), "the third velocity component must stay exactly zero"
    return a[:, :2].reshape(-1).copy()


def lin3(lin2_):
    """[cell][q][u0 u1 | g00 g01 g10 g11] -> [cell][q][u0 u1 0 | g00 g01 0  g10 g11 0  0 0 0]"""
    a = lin2_.reshape(-1, 6)
    b = np.zeros((len(a), 12))
    b[:, 0:2], b[:, 3:5], b[:, 6:8] = a[:, 0:2], a[:, 2:4], a[:, 4:6]
    return b.reshape(-1)


def lin2(lin3_):
    b = lin3_.reshape(-1, 12)
    return np.concatenate([b[:, 0:2], b[:, 3:5], b[:, 6:8]], axis=1).reshape(-1)


def q3(q2_):
    """[cell][q][2] -> [cell][q][3]"""
    return u3(q2_)


def blocks3(v2, nn):
    """block vector [2][nn] -> [3][nn]"""
    return np.concatenate([v2, np.zeros(nn)])


# ---- Navier-Stokes operators ----------------------------------------------------------------------------------------
def test_ns_operators_match_the_2d_fixture():
    """tests/golden/ns_2d_q2_8x8_newton.npz (inputs + expected outputs, no oracle call)"""
    d = load("ns_2d_q2_8x8_newton")
    p = prm_dict(d)
    k = int(d["k"])
    fp = adaflo_amd.FlowParameters(
        velocity_degree=k, physical_type=PHYS[p["physical_type"]], linearization=LIN[p["linearization"]],
        formulation_convective_term=BETA[p["beta"]], viscosity=p["viscosity"], density=p["density"],
        damping=-p["damping"], tau_grad_div=p["tau_grad_div"], density_diff=p["density_diff"])
    mesh = adaflo_amd.BrickMesh([int(n) for n in d["ncell"]], tuple(d["lower"]), tuple(d["upper"]))
    assert mesh.dim == 2
    op = adaflo_amd.NavierStokesMatrix(fp, mesh, dirichlet_faces_u=range(4), constrained_faces_p=[0])
    op.initialize(FixedTimeStepping(p), False)
    assert op.n_dofs_u() == 3 * d["src_u"].size // 2 and op.n_dofs_p() == d["src_p"].size
    op.set_linearization(lin3(d["lin"]))
    src, dst = op.block_vector(u3(d["src_u"]), d["src_p"]), op.block_vector()
    op.vmult(dst, src)
    du, dp = dst.numpy()
    assert rel_l2(u2(du), d["vmult_u"]) < TOL and rel_l2(dp, d["vmult_p"]) < TOL
    du = op.initialize_u_vector()
    op.velocity_vmult(du, src.block(0))
    assert rel_l2(u2(du.numpy()), d["velocity_vmult"]) < TOL
    dp = op.initialize_p_vector(d["src_p"])
    op.divergence_vmult_add(dp, src.block(0), False)
    assert rel_l2(dp.numpy(), d["divergence_add"]) < TOL
    op.pressure_poisson_vmult(dp, src.block(1))
    assert rel_l2(dp.numpy(), d["pressure_poisson"]) < TOL
    op.pressure_mass_vmult(dp, src.block(1))
    assert rel_l2(dp.numpy(), d["pressure_mass"]) < TOL
    res = op.block_vector()
    op.residual(res, src, None, op.block_vector(u3(d["old_u"])), op.block_vector(u3(d["oldold_u"])))
    ru, rp = res.numpy()
    assert rel_l2(u2(ru), d["residual_u"]) < TOL and rel_l2(rp, d["residual_p"]) < TOL
    assert rel_l2(lin2(op.get_linearization()), d["residual_lin"]) < TOL


@pytest.mark.parametrize("k,ncell,phys,lin,variable", [
    (2, (5, 7), 0, 0, False), (2, (4, 3), 0, 1, True), (2, (9, 4), 0, 2, False), (2, (3, 3), 0, 3, True),
    (2, (6, 5), 2, 0, False), (2, (1, 1), 0, 0, True), (3, (4, 5), 0, 0, True), (3, (3, 2), 0, 1, False),
    (3, (2, 7), 1, 0, False), (2, (40, 80), 0, 0, True)])
def test_ns_operators_equal_the_2d_oracle(k, ncell, phys, lin, variable):
    """vmult / velocity block / divergence / pressure operators / residual with its stored state, symmetry walls in x
    (normal component only) and no-slip in y as in tests/rising_bubble.cc:133-150; with and without the variable
    density / viscosity arrays of two-phase flow"""
    rng = np.random.default_rng(11 * k + ncell[0])
    lower, upper = (0.0, -0.5), (1.0, 1.5)
    omesh = orc.Mesh.make(list(ncell), lower, upper)
    mesh = adaflo_amd.BrickMesh(list(ncell), lower, upper)
    fp = adaflo_amd.FlowParameters(velocity_degree=k, physical_type=PHYS[phys], linearization=LIN[lin],
                                   formulation_convective_term="skew-symmetric", viscosity=0.07, density=1.3,
                                   tau_grad_div=0.2, density_diff=-0.6 if variable else 0.0,
                                   time_step_size_start=0.05, end_time=5.0)
    ts = adaflo_amd.TimeStepping(fp)
    ts.next(), ts.next()
    prm = orc.NSParams.make(physical_type=phys, linearization=lin, beta=0.5, tau_grad_div=0.2, density=fp.density,
                            viscosity=0.07, damping=0.0, density_diff=fp.density_diff, weight=ts.weight(),
                            weight_old=ts.weight_old(), weight_old_old=ts.weight_old_old(), tau1=ts.tau1(),
                            extrap_old=ts.factor_extrapol_old, extrap_old_old=ts.factor_extrapol_old_old)
    con_u = orc.boundary_mask(omesh, k, 2, faces=[2, 3]) | orc.boundary_mask(omesh, k, 2, faces=[0, 1], comps=[0])
    con_p = orc.boundary_mask(omesh, k - 1, 1, faces=[3])
    op = adaflo_amd.NavierStokesMatrix(fp, mesh, dirichlet_faces_u=[2, 3], symmetry_faces_u=[0, 1], constrained_faces_p=[3])
    op.initialize(ts, False)
    n_u, n_p, nq = omesh.n_nodes(k) * 2, omesh.n_nodes(k - 1), (k + 1) ** 2
    assert op.n_q_points() == nq and op.n_cells() == omesh.n_cells
    co = {}
    if variable:
        co = dict(rho=rng.uniform(0.5, 1.5, omesh.n_cells * nq), mu=rng.uniform(0.01, 0.1, omesh.n_cells * nq),
                  damp=rng.uniform(0.0, 0.3, omesh.n_cells * nq))
        op.set_coefficients(co["rho"], co["mu"], co["damp"])
    src_u, src_p = rng.uniform(-1, 1, n_u), rng.uniform(-1, 1, n_p)
    old_u, oo_u = rng.uniform(-1, 1, n_u), rng.uniform(-1, 1, n_u)
    lin_q = rng.uniform(-1, 1, omesh.n_cells * nq * 6)
    if phys != 2:
        op.set_linearization(lin3(lin_q))
    src, dst = op.block_vector(u3(src_u), src_p), op.block_vector()
    op.vmult(dst, src)
    ref_u, ref_p = orc.ns_vmult(omesh, k, prm, src_u, src_p, con_u, con_p, lin=lin_q, **co)
    du, dp = dst.numpy()
    assert rel_l2(u2(du), ref_u) < TOL and rel_l2(dp, ref_p) < TOL
    du = op.initialize_u_vector()
    op.velocity_vmult(du, src.block(0))
    assert rel_l2(u2(du.numpy()), orc.ns_velocity_vmult(omesh, k, prm, src_u, con_u, lin=lin_q, **co)) < TOL
    dp = op.initialize_p_vector(src_p)
    op.divergence_vmult_add(dp, src.block(0), False)
    assert rel_l2(dp.numpy(), orc.ns_divergence_vmult_add(omesh, k, prm, src_u, src_p, con_u, con_p)) < TOL
    if phys != 2:
        op.pressure_poisson_vmult(dp, src.block(1))
        assert rel_l2(dp.numpy(), orc.ns_pressure_poisson_vmult(omesh, k, prm, src_p, con_p, rho=co.get("rho"))) < TOL
    op.pressure_mass_vmult(dp, src.block(1))
    assert rel_l2(dp.numpy(), orc.ns_pressure_mass_vmult(omesh, k, prm, src_p, con_p, mu=co.get("mu"))) < TOL
    # residual with a user right-hand side; the state it stores drives the next vmult
    user_u = rng.uniform(-1, 1, n_u)
    lin_out = np.zeros_like(lin_q)
    ref_u, ref_p = orc.ns_residual(omesh, k, prm, src_u, src_p, old_u, oo_u, con_u=con_u, con_p=con_p, lin=lin_out,
                                   user_u=user_u, user_p=np.zeros(n_p), **co)
    res = op.block_vector()
    op.residual(res, src, op.block_vector(u3(user_u), np.zeros(n_p)), op.block_vector(u3(old_u)), op.block_vector(u3(oo_u)))
    ru, rp = res.numpy()
    assert rel_l2(u2(ru), ref_u) < TOL and rel_l2(rp, ref_p) < TOL
    if phys != 2 and lin != 3:
        op.vmult(dst, src)
        ref_u, ref_p = orc.ns_vmult(omesh, k, prm, src_u, src_p, con_u, con_p, lin=lin_out, **co)
        du, dp = dst.numpy()
        assert rel_l2(u2(du), ref_u) < TOL and rel_l2(dp, ref_p) < TOL


def test_config0_2d_beltrami_64x64_on_the_device():
    """BASELINE configs[0] (2D Q2/Q1, uniform 64 x 64 on [-1,1]^2, 33 282 + 4 225 DoF; the reference's CPU plumbing
    case, tests/test_oracle_kats.py) has a device counterpart: residual at the Beltrami interpolant, then the Newton
    vmult on the state the residual stored, against the oracle"""
    omesh = orc.Mesh.make([64, 64], [-1.0, -1.0], [1.0, 1.0])
    mesh = adaflo_amd.BrickMesh([64, 64], [-1.0, -1.0], [1.0, 1.0])
    k, dt = 2, 0.01
    fp = adaflo_amd.FlowParameters(velocity_degree=k, viscosity=1.0, density=1.0, time_step_size_start=dt, end_time=1.0)
    ts = adaflo_amd.TimeStepping(fp)
    ts.next(), ts.next()
    prm = orc.NSParams.make(beta=0.5, weight=ts.weight(), weight_old=ts.weight_old(), weight_old_old=ts.weight_old_old(),
                            tau1=ts.tau1(), extrap_old=ts.factor_extrapol_old, extrap_old_old=ts.factor_extrapol_old_old)
    op = adaflo_amd.NavierStokesMatrix(fp, mesh, dirichlet_faces_u=range(4))
    op.initialize(ts, False)
    assert (op.n_cells(), op.n_dofs_u() * 2 // 3, op.n_dofs_p()) == (4096, 33282, 4225)
    xu, xp = orc.node_coordinates(omesh, k), orc.node_coordinates(omesh, k - 1)
    rng = np.random.default_rng(7)
    # (a perturbed interpolant: the residual of the interpolant itself is the truncation error, sums that cancel to 1e-4
    # of their terms -- nothing a relative 1e-12 can be asked of)
    u, p = orc.beltrami_u(xu, 0.0).reshape(-1), orc.beltrami_p(xp, 0.0).reshape(-1)
    u = u + 0.01 * rng.uniform(-1, 1, u.size)
    u_old, u_oo = orc.beltrami_u(xu, -dt).reshape(-1), orc.beltrami_u(xu, -2 * dt).reshape(-1)
    con_u = orc.boundary_mask(omesh, k, 2)
    lin = np.zeros(omesh.n_cells * 9 * 6)
    ref_u, ref_p = orc.ns_residual(omesh, k, prm, u, p, u_old, u_oo, con_u=con_u, lin=lin)
    res = op.block_vector()
    op.residual(res, op.block_vector(u3(u), p), None, op.block_vector(u3(u_old)), op.block_vector(u3(u_oo)))
    ru, rp = res.numpy()
    assert rel_l2(u2(ru), ref_u) < TOL and rel_l2(rp, ref_p) < TOL
    du, dp = rng.uniform(-1, 1, u.size), rng.uniform(-1, 1, p.size)
    ju, jp = orc.ns_vmult(omesh, k, prm, du, dp, con_u, None, lin=lin)
    dst = op.block_vector()
    op.vmult(dst, op.block_vector(u3(du), dp))
    gu, gp = dst.numpy()
    assert rel_l2(u2(gu), ju) < TOL and rel_l2(gp, jp) < TOL


# ---- level-set operators --------------------------------------------------------------------------------------------
class LS2D:
    def __init__(self, ncell, s, k=2, faces=(), seed=5):
        lower, upper = (0.0, 0.0), (1.0, 2.0)
        self.s, self.k = s, k
        self.mesh = orc.Mesh.make(list(ncell), lower, upper)
        self.bmesh = adaflo_amd.BrickMesh(list(ncell), lower, upper)
        self.rng = np.random.default_rng(seed)
        self.nn, self.nq = self.mesh.n_nodes(s), (2 * s) ** 2
        h = [self.mesh.h[0], self.mesh.h[1]]
        self.eps_used = 1.5 * max(h) / s
        self.dt, self.weight, self.w_old, self.w_oo, self.epsilon = 0.02, 75.0, -100.0, 25.0, 1.5
        self.prm = orc.make_ls_params(s, self.eps_used, min(h), self.dt, self.weight, max(h), self.epsilon)
        self.con = orc.boundary_mask(self.mesh, s, 1, faces=list(faces)) if faces else None
        self.ops = lso.LevelSetOperators(self.bmesh, s, velocity_degree=k, constrained_faces=faces)
        assert self.ops.n_dofs == self.nn and self.ops.n_q == self.nq
        self.ops.set_parameters(self.eps_used, self.dt, self.weight, self.w_old, self.w_oo, self.epsilon)
        self.diag = self.rng.uniform(0.5, 2.0, self.nn)
        if faces:
            self.ops.set_diagonal(self.ops.vector(self.diag))

    def rand(self, blocks=1):
        return self.rng.uniform(-1, 1, self.nn * blocks)

    def rand_q(self):
        return self.rng.uniform(-1, 1, self.mesh.n_cells * self.nq * 2)


@pytest.mark.parametrize("s,ncell,faces", [(4, (3, 2), ()), (2, (5, 4), (0, 3)), (1, (7, 6), ()), (3, (2, 3), (2,)),
                                           (4, (40, 80), ()), (3, (1, 1), (1,)), (2, (33, 9), (0, 1, 2, 3))])
def test_ls_operators_equal_the_2d_oracle(s, ncell, faces):
    c = LS2D(ncell, s, faces=faces)
    nn = c.nn
    src = c.rand()
    d = c.ops.vector(np.full(nn, 9.0))
    adv = lso.LevelSetOKZSolverAdvanceConcentration(c.ops)
    uq = c.rand_q()
    adv.evaluated_convection = q3(uq)
    adv.advance_concentration_vmult(d, c.ops.vector(src))
    assert rel_l2(d.numpy(), orc.ls_advect_vmult(c.mesh, c.prm, src, uq, con=c.con, diag=c.diag)) < TOL
    rei = lso.LevelSetOKZSolverReinitialization(c.ops)
    nq = c.rand_q()
    rei.evaluated_normal = q3(nq)
    for diffuse_only in (False, True):
        rei.reinitialization_vmult(d, c.ops.vector(src), diffuse_only)
        ref = orc.ls_reinit_vmult(c.mesh, c.prm, src, nq, diffuse_only=diffuse_only, con=c.con, diag=c.diag)
        assert rel_l2(d.numpy(), ref) < TOL
    src2 = c.rand(2)
    d3 = c.ops.vector(blocks=3)
    lso.LevelSetOKZSolverComputeNormal(c.ops).compute_normal_vmult(d3, c.ops.vector(blocks3(src2, nn), blocks=3))
    out = d3.numpy()
    assert rel_l2(out[:2 * nn], orc.ls_normal_vmult(c.mesh, c.prm, src2, con=c.con, diag=c.diag)) < TOL
    assert np.all(out[2 * nn:] == 0.0)
    cur = lso.LevelSetOKZSolverComputeCurvature(c.ops)
    for apply_diffusion in (True, False):
        cur.compute_curvature_vmult(d, c.ops.vector(src), apply_diffusion)
        ref = orc.ls_curvature_vmult(c.mesh, c.prm, src, apply_diffusion=apply_diffusion, con=c.con, diag=c.diag)
        assert rel_l2(d.numpy(), ref) < TOL
    # ---- right-hand sides
    phi = c.rand()
    normal = c.rand(2)
    normal[::7] *= 1e-3
    nq_ref = np.zeros(c.mesh.n_cells * c.nq * 2)
    ref = orc.ls_reinit_rhs(c.mesh, c.prm, phi, normal, nq_ref, diffuse_only=False, first_step=True, con=c.con)
    d = c.ops.vector()
    rei.local_reinitialize_rhs(d, c.ops.vector(phi), c.ops.vector(blocks3(normal, nn), blocks=3), False, True)
    assert rel_l2(d.numpy(), ref) < TOL
    assert rel_l2(rei.evaluated_normal, q3(nq_ref)) < TOL
    phi2 = c.rand()
    for diffuse_only in (False, True):
        ref = orc.ls_reinit_rhs(c.mesh, c.prm, phi2, normal, nq_ref, diffuse_only=diffuse_only, first_step=False, con=c.con)
        d = c.ops.vector()
        rei.local_reinitialize_rhs(d, c.ops.vector(phi2), None, diffuse_only, False)
        assert rel_l2(d.numpy(), ref) < TOL
    d3 = c.ops.vector(blocks=3)
    lso.LevelSetOKZSolverComputeNormal(c.ops).local_compute_normal_rhs(d3, c.ops.vector(phi))
    out = d3.numpy()
    assert rel_l2(out[:2 * nn], orc.ls_normal_rhs(c.mesh, c.prm, phi, con=c.con)) < TOL and np.all(out[2 * nn:] == 0.0)
    normal_z = normal.reshape(2, -1).copy()
    normal_z[:, : nn // 3] = 0.0
    normal_z = normal_z.reshape(-1)
    d = c.ops.vector()
    cur.local_compute_curvature_rhs(d, c.ops.vector(blocks3(normal_z, nn), blocks=3))
    assert rel_l2(d.numpy(), orc.ls_curvature_rhs(c.mesh, c.prm, normal_z, con=c.con)) < TOL
    # advection right-hand side (BDF-2 history, velocity evaluated at the level-set quadrature points)
    k = c.k
    vel = c.rng.uniform(-1, 1, c.mesh.n_nodes(k) * 2)
    old, oo = c.rand(), c.rand()
    vq_ref = np.zeros(c.mesh.n_cells * c.nq * 2)
    ref = orc.ls_advect_rhs(c.mesh, c.prm, k, phi, old, oo, vel, vq_ref, c.w_old, c.w_oo, use_old_old=True, con=c.con)
    d = c.ops.vector()
    adv.local_advance_concentration_rhs(d, c.ops.vector(phi), c.ops.vector(old), c.ops.vector(oo),
                                        c.ops.velocity_vector(u3(vel)), True)
    assert rel_l2(d.numpy(), ref) < TOL
    assert rel_l2(adv.evaluated_convection, q3(vq_ref)) < TOL
    assert abs(adv.get_maximal_velocity(c.ops.velocity_vector(u3(vel))) - orc.ls_max_velocity(c.mesh, k, vel)) < 1e-13
    # mass diagonal of the DiagonalPreconditioner (unconstrained spaces only: the constrained rows hold the diagonal)
    if not faces:
        idx = np.indices((s * ncell[1] + 1, s * ncell[0] + 1))
        col = ((idx[0] % 2) * 2 + idx[1] % 2).reshape(-1)
        diag = np.zeros(nn)
        for colour in range(4):
            y = orc.ls_curvature_vmult(c.mesh, c.prm, (col == colour).astype(float), apply_diffusion=False)
            diag[col == colour] = y[col == colour]
        got = c.ops.initialize_mass_matrix_diagonal().diagonal_vector.numpy()
        assert rel_l2(got, diag) < TOL


@pytest.mark.parametrize("k,s,ncell,variable,interpolate", [(2, 4, (6, 12), True, True), (2, 3, (5, 10), False, True),
                                                             (3, 2, (4, 8), True, True), (2, 2, (7, 5), True, False),
                                                             (2, 4, (40, 80), True, True)])
def test_heaviside_and_force_equal_the_2d_oracle(k, s, ncell, variable, interpolate):
    """compute_heaviside + compute_force (level_set_okz.cc:317-540) in 2D: gravity acts along y, the density /
    viscosity arrays have (k+1)^2 points per cell"""
    lower, upper = (0.0, 0.0), (1.0, 2.0)
    omesh = orc.Mesh.make(list(ncell), lower, upper)
    mesh = adaflo_amd.BrickMesh(list(ncell), lower, upper)
    phys = dict(surface_tension=0.0245, gravity=0.98, density=1.0, density_diff=-0.9 if variable else 0.0,
                viscosity=0.01, viscosity_diff=-0.009 if variable else 0.0)
    fp = adaflo_amd.FlowParameters(velocity_degree=k, concentration_subdivisions=s, epsilon=1.5,
                                   interpolate_grad_onto_pressure=interpolate, time_step_size_start=0.02, end_time=1.0,
                                   **phys)
    ts = adaflo_amd.TimeStepping(fp)
    ts.next()
    op = adaflo_amd.NavierStokesMatrix(fp, mesh, dirichlet_faces_u=[2, 3], symmetry_faces_u=[0, 1], ls_degree=s)
    op.initialize(ts, True)
    ops = lso.LevelSetOperators(mesh, s, velocity_degree=k, navier_stokes_matrix=op)
    x = orc.node_coordinates(omesh, s, fe_type=1)
    eps_used = 1.5 / s * omesh.h[0]
    phi = -np.tanh((np.linalg.norm(x - np.array([0.45, 0.6]), axis=1) - 0.27) / (2 * eps_used))
    rng = np.random.default_rng(3)
    kappa = 1.0 / 0.27 + 0.1 * rng.uniform(-1, 1, phi.size)
    hv = ops.vector()
    ops.compute_heaviside(hv, ops.vector(phi), 1.5)
    H = orc.ls_compute_heaviside(omesh, s, 1.5, phi)
    assert rel_l2(hv.numpy(), H) < TOL
    con_u = orc.boundary_mask(omesh, k, 2, faces=[2, 3]) | orc.boundary_mask(omesh, k, 2, faces=[0, 1], comps=[0])
    ref, rho, mu = orc.ls_compute_force(omesh, s, k, H, kappa, interpolate_grad_onto_pressure=interpolate, con_u=con_u,
                                        **phys)
    f = op.initialize_u_vector()
    ops.compute_force(f, hv, ops.vector(kappa), fp)
    assert rel_l2(u2(f.numpy()), ref) < TOL
    if variable:
        got_rho, got_mu, _ = op.get_coefficients()
        assert rel_l2(got_rho, rho) < TOL and rel_l2(got_mu, mu) < TOL


# ---- the reference's printed numbers ----------------------------------------------------------------------------------
def _bubble_parameters(ref, k=2, s=4, linearization="coupled implicit Newton", max_nl=10, dt=0.02, **physics):
    phys = dict(density=1.0, density_diff=-0.9, viscosity=0.01, viscosity_diff=-0.009, surface_tension=0.0245, gravity=0.98)
    phys.update(physics)
    # tests/rising_bubble_ls.prm: NL tolerance 1e-9, lin its 30 relative 1e-4 (the device's FGMRES + block
    # preconditioner with Jacobi-type inner solves gets more iterations: it has no ILU / AMG)
    return adaflo_amd.FlowParameters(
        velocity_degree=k, epsilon=1.5, concentration_subdivisions=s, interpolate_grad_onto_pressure=True,
        curvature_correction=True, time_step_size_start=dt, end_time=1.0, linearization=linearization,
        max_nl_iteration=max_nl, tol_nl_iteration=1e-9, max_lin_iteration=500, tol_lin_iteration=1e-4, **phys)


def _as_oracle_sim(dev, ncell, physics=None):
    """the device solution in the oracle's 2D layout, for the host-side statistics of the reference
    (two_phase_base.cc:621-905, tests/spurious_currents.cc:121-222), which the oracle module restates"""
    m = dev.mesh
    return types.SimpleNamespace(
        dim=2, mesh=orc.Mesh.make(list(ncell), tuple(m.lower[:2]), tuple(m.upper[:2])), s=dev.parameters.concentration_subdivisions,
        k=dev.parameters.velocity_degree, ncell=list(ncell), phi=dev.solution.numpy(),
        u=u2(dev.navier_stokes.solution[0].cpu().numpy()), p=dev.navier_stokes.solution[1].cpu().numpy(),
        physics=physics or {})


def _check_steps(dev, ref, ncell, statistics="bubble", physics=None):
    from oracle import two_phase_oracle as tpo
    for no, expected in enumerate(ref["time_steps"]):
        dev.navier_stokes.history.clear()
        dev.advance_time_step()
        adv_it, adv_r0 = dev.concentration_iterations[-1]
        assert adv_it == expected["advect_iterations"], (no, adv_it, adv_r0)
        if expected["advect_residual"] == "0":
            assert adv_r0 < 1e-12
        else:
            assert "%.3g" % adv_r0 == expected["advect_residual"], (no, adv_r0)
        assert dev.reinit_iterations[-1] == expected["reinitialize_iterations"], (no, dev.reinit_iterations[-1])
        history = [float(np.hypot(*h)) for h in dev.navier_stokes.history]
        assert "%.3g" % history[0] == expected["first_residual"], (no, history)
        if dev.parameters.max_nl_iteration > 1:
            assert history[-1] < 1e-9, (no, history)
        sim = _as_oracle_sim(dev, ncell, physics)
        if statistics == "bubble":
            circ, vel, centre, _ = tpo.bubble_statistics_2d(sim)
            mine = dev.compute_bubble_statistics()                  # the product's own evaluation prints the same lines
            assert abs(mine["circularity"] - circ) < 1e-12 and np.allclose(mine["velocity"], vel, rtol=1e-11, atol=1e-16)
            assert np.allclose(mine["centre"], centre, rtol=1e-12)
            assert mine["lines"][0].startswith("  Degree of circularity: " + expected["circularity"][:8])
            assert abs(circ - float(expected["circularity"])) < 1.5e-8, (no, circ)
            # (three units of the last printed digit: both codes stop their Newton iteration at a residual of 1e-9, the
            # reference with its ILU-preconditioned linear solves, the device with Jacobi-type inner solves)
            assert abs(vel[0]) < 1e-7 * abs(vel[1]) and abs(vel[1] - float(expected["mean_bubble_velocity_y"])) < 3e-9, (no, vel)
            assert abs(centre[0] - 0.5) < 1e-9 and abs(centre[1] - float(expected["centre_of_mass_y"])) < 1.5e-8, (no, centre)
        else:
            jump, size = tpo.spurious_current_statistics_2d(sim)
            assert abs(jump - float(expected["pressure_jump_error_percent"])) < 1e-6, (no, jump)
            assert abs(size - float(expected["size_spurious_currents"])) < 1e-6 * size, (no, size)


def test_rising_bubble_prints_the_reference_output():
    """tests/rising_bubble_ls.output on the device: 40 x 80 cells, Q2/Q1 + FE_Q_iso_Q1(4)
        reinitialize (8 + 8)
        step 1   advect [0/0]        reinitialize (7 + 7)     first residual 0.0198
        step 2   advect [0.000471/9] reinitialize (11 + 10)   first residual 0.00581
        step 3   advect [0.00108/10] reinitialize (11 + 11)   first residual 0.000246
    and circularity / mean bubble velocity / centre of mass to the printed 8 digits after every step"""
    with open(GOLDEN) as f:
        ref = json.load(f)["rising_bubble_ls"]
    mesh = adaflo_amd.BrickMesh([40, 80], [0., 0.], [1., 2.])
    dev = LevelSetOKZSolver(_bubble_parameters(ref), mesh, lambda x: np.linalg.norm(x[:, :2] - 0.5, axis=1) - 0.25,
                            symmetry_faces=[0, 1])
    m = dev.navier_stokes.navier_stokes_matrix
    assert (m.n_cells(), m.n_dofs_u() * 2 // 3, m.n_dofs_p(), dev.ops.n_dofs) == (
        ref["cells"], ref["dofs_u"], ref["dofs_p"], ref["dofs_ls"])
    assert dev.initial_reinit_iterations == ref["initial_reinitialize_iterations"]
    _check_steps(dev, ref, (40, 80))


@pytest.mark.parametrize("case,linearization", [("rising_bubble_ls_picard", "coupled implicit Picard"),
                                                ("rising_bubble_ls_imex", "coupled velocity semi-implicit"),
                                                ("rising_bubble_ls_expl", "coupled velocity explicit")])
def test_rising_bubble_other_linearisations_print_their_reference_outputs(case, linearization):
    """tests/rising_bubble_ls_{picard,imex,expl}.output: FE_Q_iso_Q1(3); the three runs differ in the first residual of
    step #3 (0.000244 / 0.000245 / 0.000246)"""
    with open(GOLDEN) as f:
        ref = json.load(f)[case]
    mesh = adaflo_amd.BrickMesh([40, 80], [0., 0.], [1., 2.])
    fp = _bubble_parameters(ref, s=ref["concentration_subdivisions"], linearization=linearization, max_nl=ref["nl_max_iterations"])
    dev = LevelSetOKZSolver(fp, mesh, lambda x: np.linalg.norm(x[:, :2] - 0.5, axis=1) - 0.25, symmetry_faces=[0, 1])
    assert dev.ops.n_dofs == ref["dofs_ls"]
    assert dev.initial_reinit_iterations == ref["initial_reinitialize_iterations"]
    _check_steps(dev, ref, (40, 80))


def test_rising_bubble_q3_q2_prints_its_reference_output():
    """tests/rising_bubble_ls_q3.output: Taylor-Hood Q3/Q2 on 20 x 40 cells"""
    with open(GOLDEN) as f:
        ref = json.load(f)["rising_bubble_ls_q3"]
    mesh = adaflo_amd.BrickMesh([20, 40], [0., 0.], [1., 2.])
    fp = _bubble_parameters(ref, k=ref["velocity_degree"], s=ref["concentration_subdivisions"])
    dev = LevelSetOKZSolver(fp, mesh, lambda x: np.linalg.norm(x[:, :2] - 0.5, axis=1) - 0.25, symmetry_faces=[0, 1])
    m = dev.navier_stokes.navier_stokes_matrix
    assert (m.n_cells(), m.n_dofs_u() * 2 // 3, m.n_dofs_p(), dev.ops.n_dofs) == (
        ref["cells"], ref["dofs_u"], ref["dofs_p"], ref["dofs_ls"])
    assert dev.initial_reinit_iterations == ref["initial_reinitialize_iterations"]
    _check_steps(dev, ref, (20, 40))


def test_spurious_currents_prints_its_reference_output():
    """tests/spurious_currents_ls.output: static bubble of radius 0.5 at (0.02, 0.03) in [-2.5, 2.5]^2, 80 x 80 cells,
    no-slip walls, equal densities / viscosities, sigma = 1, FE_Q_iso_Q1(3), no initial reinitialisation, dt = 0.01"""
    with open(GOLDEN) as f:
        ref = json.load(f)["spurious_currents_ls"]
    lower, upper = ref["domain"]
    mesh = adaflo_amd.BrickMesh([80, 80], lower, upper)
    ph = ref["physics"]
    fp = _bubble_parameters(ref, s=ref["concentration_subdivisions"], dt=ref["dt"], density=ph["density"],
                            density_diff=ph["density_diff"], viscosity=ph["viscosity"], viscosity_diff=ph["viscosity_diff"],
                            surface_tension=ph["surface_tension"], gravity=ph["gravity"])
    centre = np.asarray(ref["centre"])
    dev = LevelSetOKZSolver(fp, mesh, lambda x: np.linalg.norm(x[:, :2] - centre, axis=1) - ref["radius"],
                            n_initial_reinit_steps=0)
    m = dev.navier_stokes.navier_stokes_matrix
    assert (m.n_cells(), m.n_dofs_u() * 2 // 3, m.n_dofs_p(), dev.ops.n_dofs) == (
        ref["cells"], ref["dofs_u"], ref["dofs_p"], ref["dofs_ls"])
    _check_steps(dev, ref, (80, 80), statistics="spurious", physics=ph)


@pytest.mark.parametrize("refinements", [4, 6])
def test_poiseuille_stokes_prints_its_reference_output(refinements):
    """tests/poiseuille_stokes.output (tests/poiseuille.cc, poiseuille_stokes.prm; 6 global refinements = 256 x 64 cells,
    132 354 + 16 705 DoF): Stokes flow in the half channel [-2, 2] x [-1, 0], no-slip wall at the bottom, symmetry at
    y = 0, open boundaries with normal flux and the pressure 2 - x at both ends.  The first residual -- the open-boundary
    face integrals on the free rows -- prints `3.722e-01`; the exact solution u = (1 - y^2) / (2 nu), p = 2 - x lies in the
    Q2/Q1 space, so the converged errors are round-off (the reference prints 9.5e-12 / 5.4e-12)."""
    nu = 0.1
    n = 2 ** refinements
    mesh = adaflo_amd.BrickMesh([4 * n, n], [-2.0, -1.0], [2.0, 0.0])
    fp = adaflo_amd.FlowParameters(velocity_degree=2, physical_type="stokes", viscosity=nu, time_step_size_start=0.01,
                                   end_time=1.0, max_nl_iteration=10, tol_nl_iteration=1e-10, max_lin_iteration=500,
                                   tol_lin_iteration=1e-5)
    from adaflo_amd.navier_stokes import NavierStokes, node_coordinates
    p_ext = lambda x, t: 2.0 - x[:, 0]
    ns = NavierStokes(fp, mesh, adaflo_amd.TimeStepping(fp), dirichlet_function=lambda x, t: np.zeros((len(x), 2)),
                      symmetry_faces=[3], open_faces={0: p_ext, 1: p_ext})
    m = ns.navier_stokes_matrix
    if refinements == 6:
        assert (m.n_cells(), m.n_dofs_u() * 2 // 3, m.n_dofs_p()) == (16384, 132354, 16705)   # poiseuille_stokes.output:2-3
    ns.init_time_advance()
    res = ns.compute_residual()
    if refinements == 6:
        assert "%.3e" % res == _ref("poiseuille_stokes")["first_residual"]                                                   # poiseuille_stokes.output:11
    ns.solve_nonlinear_system(res)
    assert np.hypot(*ns.history[-1]) < 1e-10, ns.history
    xu, xp = node_coordinates(mesh, 2), node_coordinates(mesh, 1)
    u = ns.solution[0].cpu().numpy().reshape(-1, 3)
    p = ns.solution[1].cpu().numpy()
    assert np.abs(u[:, 0] - 0.5 / nu * (1 - xu[:, 1] ** 2)).max() < 1e-8 and np.abs(u[:, 1:]).max() < 1e-8
    assert np.abs(p - (2.0 - xp[:, 0])).max() < 1e-8


def test_poiseuille_navier_stokes_prints_its_reference_output():
    """tests/poiseuille_ns.output (poiseuille_ns.prm: 64 x 16 cells, 8 514 + 1 105 DoF, nu = 0.5, BDF-2 with dt = 0.5 from
    rest): the channel flow develops towards the parabolic profile.  The first nonlinear residual of every time step
    depends only on the converged steps before it: `7.419e-01`, `5.800e-02`, `2.307e-02`, `1.560e-02` (:11, :31, :40,
    :49); after four steps the velocity error against the STEADY profile prints `0.1321` (:56, QGauss(4) per cell)."""
    from adaflo_amd.navier_stokes import NavierStokes, node_coordinates
    from common import l2_norm_of_difference
    nu = 0.5
    mesh = adaflo_amd.BrickMesh([64, 16], [-2.0, -1.0], [2.0, 0.0])
    fp = adaflo_amd.FlowParameters(velocity_degree=2, viscosity=nu, time_step_size_start=0.5, end_time=20.0,
                                   max_nl_iteration=10, tol_nl_iteration=1e-11, max_lin_iteration=500, tol_lin_iteration=1e-5)
    p_ext = lambda x, t: 2.0 - x[:, 0]
    ns = NavierStokes(fp, mesh, adaflo_amd.TimeStepping(fp), dirichlet_function=lambda x, t: np.zeros((len(x), 2)),
                      symmetry_faces=[3], open_faces={0: p_ext, 1: p_ext})
    m = ns.navier_stokes_matrix
    assert (m.n_cells(), m.n_dofs_u() * 2 // 3, m.n_dofs_p()) == (1024, 8514, 1105)           # poiseuille_ns.output:2-3
    for expected in _ref("poiseuille_ns")["first_residuals"]:
        ns.history.clear()
        ns.advance_time_step()
        assert "%.3e" % np.hypot(*ns.history[0]) == expected, ns.history
        assert np.hypot(*ns.history[-1]) < 1e-11, ns.history
    # ||e_u||_L2 against u = (1 - y^2) / (2 nu) e_x with QGauss(k + 2) (tests/poiseuille.cc:150-190)
    omesh = orc.Mesh.make([64, 16], (-2.0, -1.0), (2.0, 0.0))
    u = u2(ns.solution[0].cpu().numpy())
    xq, wq = orc.gauss_legendre(4)
    S, _ = orc.shape_1d(0, 2, xq)
    uu = u.reshape(33, 129, 2)
    iy = np.arange(16)[:, None] * 2 + np.arange(3)[None, :]
    ix = np.arange(64)[:, None] * 2 + np.arange(3)[None, :]
    loc = uu[iy[:, None, :, None], ix[None, :, None, :]]                                    # [cy][cx][j][i][c]
    val = np.einsum("qj,pi,yxjic->yxqpc", S, S, loc)
    yq = -1.0 + (np.arange(16)[:, None] + xq[None, :]) / 16.0                                 # [cy][q]
    exact = np.zeros_like(val)
    exact[..., 0] = (0.5 / nu * (1 - yq ** 2))[:, None, :, None]
    w = np.outer(wq, wq) / 16.0 / 16.0
    err = np.sqrt(np.einsum("yxqpc,qp->", (val - exact) ** 2, w))
    assert "%.4g" % err == _ref("poiseuille_ns")["l2_error_u_after_four_steps"], err


def test_couette_prints_its_reference_output():
    """tests/couette.output (tests/couette.cc, couette.prm: 64 x 16 cells, nu = 0.5, BDF-2 with dt = 0.5): the top wall
    moves with velocity (2, 0), the bottom wall rests, both ends are open (normal flux, zero pressure).  First nonlinear
    residuals of the two time steps: `1.601e+01` (:10) and `1.930e-01` (:31)."""
    from adaflo_amd.navier_stokes import NavierStokes
    mesh = adaflo_amd.BrickMesh([64, 16], [-2.0, -1.0], [2.0, 0.0])
    fp = adaflo_amd.FlowParameters(velocity_degree=2, viscosity=0.5, time_step_size_start=0.5, end_time=1.0,
                                   max_nl_iteration=10, tol_nl_iteration=1e-11, max_lin_iteration=500, tol_lin_iteration=1e-5)
    wall = lambda x, t: np.stack([np.where(np.abs(x[:, 1]) < 1e-13, 2.0, 0.0), np.zeros(len(x))], axis=1)
    zero = lambda x, t: np.zeros(len(x))
    ns = NavierStokes(fp, mesh, adaflo_amd.TimeStepping(fp), dirichlet_function=wall, open_faces={0: zero, 1: zero})
    for expected in _ref("couette")["first_residuals"]:
        ns.history.clear()
        ns.advance_time_step()
        assert "%.3e" % np.hypot(*ns.history[0]) == expected, ns.history
        assert np.hypot(*ns.history[-1]) < 1e-11, ns.history
    # the steady state is the linear profile u = 2 (1 + y): after two steps the flow is on its way there
    u = ns.solution[0].cpu().numpy().reshape(33, 129, 3)
    assert np.all(np.diff(u[:, 64, 0]) > 0) and abs(u[-1, 64, 0] - 2.0) < 1e-14 and np.abs(u[:, :, 1]).max() < 1e-9


@pytest.mark.parametrize("case", ["1d_flow", "1d_flow_damped"])
def test_one_dimensional_flows_print_their_reference_outputs(case):
    """dim = 1 on the device (NavierStokesMatrix<1>, navier_stokes_matrix.cc:1210: both transverse directions flat):
    tests/1d_flow.output:10,30 and tests/1d_flow_damped.output:10,30,39,47,55 -- [0, 2.5] with 2048 cells, u = 2 at
    t = 0, open ends with the pressures 2 and 1, tau grad div = 1e-5; the damped run is the only reference output that
    exercises the damping term of the operator (:831-835)"""
    from adaflo_amd.navier_stokes import NavierStokes
    ref = _ref(case)
    mesh = adaflo_amd.BrickMesh([ref["cells"]], [0.0], [2.5])
    fp = adaflo_amd.FlowParameters(velocity_degree=2, viscosity=ref["viscosity"], damping=ref["damping"],
                                   tau_grad_div=ref["tau_grad_div"], time_step_size_start=ref["dt"], end_time=1.0,
                                   max_nl_iteration=10, tol_nl_iteration=1e-11, max_lin_iteration=500, tol_lin_iteration=1e-5)
    ns = NavierStokes(fp, mesh, adaflo_amd.TimeStepping(fp), dirichlet_function=lambda x, t: np.zeros((len(x), 1)),
                      open_faces={0: lambda x, t: np.full(len(x), 2.0), 1: lambda x, t: np.full(len(x), 1.0)})
    m = ns.navier_stokes_matrix
    assert (m.n_cells(), m.n_dofs_u() // 3, m.n_dofs_p()) == (ref["cells"], ref["dofs_u"], ref["dofs_p"])
    u0 = np.zeros((ref["dofs_u"], 3))
    u0[:, 0] = 2.0
    ns.set_initial_condition(u0.reshape(-1), np.zeros(ref["dofs_p"]))
    for expected in ref["first_residuals"]:
        ns.history.clear()
        ns.advance_time_step()
        assert "%.3e" % np.hypot(*ns.history[0]) == expected, (case, ns.history)
        assert np.hypot(*ns.history[-1]) < 1e-11, (case, ns.history)
    u = ns.solution[0].cpu().numpy().reshape(-1, 3)
    assert np.abs(u[:, 0] - u[0, 0]).max() < 1e-9 and np.all(u[:, 1:] == 0.0)      # incompressible in 1D: uniform


@pytest.mark.parametrize("k,n,phys,lin", [(2, 37, 0, 0), (3, 16, 0, 1), (2, 1, 0, 0), (2, 64, 2, 0), (3, 9, 1, 0)])
def test_ns_operators_equal_the_1d_oracle(k, n, phys, lin):
    """dim = 1: vmult / velocity block / divergence / pressure operators / residual with its stored state against the
    oracle's one-dimensional operators (damping, tau grad div, an open and a Dirichlet end)"""
    rng = np.random.default_rng(5 * k + n)
    omesh = orc.Mesh.make([n], (0.3,), (2.8,))
    mesh = adaflo_amd.BrickMesh([n], [0.3], [2.8])
    fp = adaflo_amd.FlowParameters(velocity_degree=k, physical_type=PHYS[phys], linearization=LIN[lin], viscosity=0.07,
                                   density=1.3, damping=0.2, tau_grad_div=0.3, time_step_size_start=0.05, end_time=5.0)
    ts = adaflo_amd.TimeStepping(fp)
    ts.next(), ts.next()
    prm = orc.NSParams.make(physical_type=phys, linearization=lin, beta=0.5, tau_grad_div=0.3, density=fp.density,
                            viscosity=0.07, damping=-0.2, weight=ts.weight(), weight_old=ts.weight_old(),
                            weight_old_old=ts.weight_old_old(), tau1=ts.tau1(), extrap_old=ts.factor_extrapol_old,
                            extrap_old_old=ts.factor_extrapol_old_old)
    con_u = orc.boundary_mask(omesh, k, 1, faces=[1])
    op = adaflo_amd.NavierStokesMatrix(fp, mesh, dirichlet_faces_u=[1])
    op.initialize(ts, False)
    n_u, n_p, nq = omesh.n_nodes(k), omesh.n_nodes(k - 1), k + 1
    assert op.n_q_points() == nq and op.n_dofs_u() == 3 * n_u and op.n_dofs_p() == n_p

    def u3_(u1):
        a = np.zeros((u1.size, 3))
        a[:, 0] = u1
        return a.reshape(-1)

    def u1_(u3v):
        a = np.asarray(u3v).reshape(-1, 3)
        assert np.all(a[:, 1:] == 0.0)
        return a[:, 0].copy()

    def lin3_(l1):                                              # [u | du/dx] -> the 12-double record
        b = np.zeros((l1.size // 2, 12))
        b[:, 0], b[:, 3] = l1.reshape(-1, 2)[:, 0], l1.reshape(-1, 2)[:, 1]
        return b.reshape(-1)
    src_u, src_p = rng.uniform(-1, 1, n_u), rng.uniform(-1, 1, n_p)
    old_u, oo_u = rng.uniform(-1, 1, n_u), rng.uniform(-1, 1, n_u)
    lin_q = rng.uniform(-1, 1, omesh.n_cells * nq * 2)
    if phys != 2:
        op.set_linearization(lin3_(lin_q))
    src, dst = op.block_vector(u3_(src_u), src_p), op.block_vector()
    op.vmult(dst, src)
    ref_u, ref_p = orc.ns_vmult(omesh, k, prm, src_u, src_p, con_u, None, lin=lin_q)
    du, dp = dst.numpy()
    assert rel_l2(u1_(du), ref_u) < TOL and rel_l2(dp, ref_p) < TOL
    du = op.initialize_u_vector()
    op.velocity_vmult(du, src.block(0))
    assert rel_l2(u1_(du.numpy()), orc.ns_velocity_vmult(omesh, k, prm, src_u, con_u, lin=lin_q)) < TOL
    dp = op.initialize_p_vector(src_p)
    op.divergence_vmult_add(dp, src.block(0), False)
    assert rel_l2(dp.numpy(), orc.ns_divergence_vmult_add(omesh, k, prm, src_u, src_p, con_u, None)) < TOL
    if phys != 2:
        op.pressure_poisson_vmult(dp, src.block(1))
        assert rel_l2(dp.numpy(), orc.ns_pressure_poisson_vmult(omesh, k, prm, src_p, None)) < TOL
    op.pressure_mass_vmult(dp, src.block(1))
    assert rel_l2(dp.numpy(), orc.ns_pressure_mass_vmult(omesh, k, prm, src_p, None)) < TOL
    lin_out = np.zeros_like(lin_q)
    ref_u, ref_p = orc.ns_residual(omesh, k, prm, src_u, src_p, old_u, oo_u, con_u=con_u, lin=lin_out)
    res = op.block_vector()
    op.residual(res, src, None, op.block_vector(u3_(old_u)), op.block_vector(u3_(oo_u)))
    ru, rp = res.numpy()
    assert rel_l2(u1_(ru), ref_u) < TOL and rel_l2(rp, ref_p) < TOL
    if phys != 2:
        got = op.get_linearization().reshape(-1, 12)
        # (Picard-type states hold (u, div u); in 1D div u = du/dx, so slot 3 is right for both schemes)
        assert rel_l2(got[:, [0, 3]], lin_out.reshape(-1, 2)) < TOL

"""GPU parity tests: HIP engine (through the C ABI) vs the CPU oracle on identical
seeded inputs.  Tolerance from BASELINE.json north_star: 1e-12 relative L2."""
import numpy as np
import pytest

from common import Case, rel_l2
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-12


def run_vmult(case, variant=0, coefficients=False):
    src_u, src_p = case.random_u(), case.random_p()
    lin = case.random_lin()
    coef = case.random_coefficients() if coefficients else (None, None, None)
    w, modes = case.weights_modes()
    ref_u, ref_p = orc.ns_vmult(case.mesh, case.k, case.prm, src_u, src_p, case.con_u, case.con_p,
                                lin=lin, rho=coef[0], mu=coef[1], damp=coef[2], weights=w, modes=modes)
    op = case.engine()
    op.set_kernel_variant(variant)
    op.set_linearization(lin)
    if coefficients:
        op.set_coefficients(*coef)
    src = op.block_vector(src_u, src_p)
    dst = op.block_vector(np.full(case.n_u, 7.0), np.full(case.n_p, -3.0))  # must be overwritten
    op.vmult(dst, src)
    got_u, got_p = dst.numpy()
    return rel_l2(got_u, ref_u), rel_l2(got_p, ref_p)


@pytest.mark.parametrize("k,ncell", [(2, (4, 3, 5)), (3, (3, 2, 2)), (4, (2, 2, 3))])
def test_vmult_newton_generic(k, ncell):
    case = Case(ncell, k=k, upper=(1.0, 0.5, 2.0))
    eu, ep = run_vmult(case, variant=0)
    assert eu < TOL and ep < TOL, (eu, ep)


@pytest.mark.parametrize("lin", [1, 2, 3, 4])
def test_vmult_linearizations_generic(lin):
    case = Case((3, 4, 2), k=2, linearization=lin, beta=1.0, tau_grad_div=0.3)
    eu, ep = run_vmult(case, variant=0)
    assert eu < TOL and ep < TOL, (eu, ep)


def test_velocity_degree_six():
    """degree_p = 5, the last instance of EXPAND_OPERATIONS (navier_stokes_matrix.cc:64-82): vmult, residual with the
    state it stores and velocity_vmult of the Q6/Q5 pair on the generic kernels (sub-blocks: test_scalar_sub_blocks)"""
    case = Case((2, 2, 3), k=6, upper=(1.0, 0.5, 2.0), tau_grad_div=0.2, steps=3)
    eu, ep = run_vmult(case, variant=1)
    assert eu < TOL and ep < TOL, (eu, ep)
    src_u, src_p = case.smooth_u(0.1) + 0.01 * case.random_u(), case.smooth_p(0.1)
    old_u, oldold_u = case.smooth_u(0.05), case.smooth_u(0.0)
    lin_ref = np.zeros(case.n_cells * case.nq * 12)
    ref_u, ref_p = orc.ns_residual(case.mesh, case.k, case.prm, src_u, src_p, old_u, oldold_u,
                                   con_u=case.con_u, con_p=case.con_p, lin=lin_ref)
    op = case.engine()
    rhs = op.block_vector()
    op.residual(rhs, op.block_vector(src_u, src_p), None, op.block_vector(old_u), op.block_vector(oldold_u))
    got_u, got_p = rhs.numpy()
    assert rel_l2(got_u, ref_u) < TOL and rel_l2(got_p, ref_p) < TOL
    assert rel_l2(op.get_linearization(), lin_ref) < TOL
    su = case.random_u()
    ref = orc.ns_velocity_vmult(case.mesh, case.k, case.prm, su, case.con_u, lin=lin_ref)
    op.fix_linearization_point()
    dst = op.initialize_u_vector()
    op.velocity_vmult(dst, op.initialize_u_vector(su))
    assert rel_l2(dst.numpy(), ref) < TOL


@pytest.mark.parametrize("phys", [1, 2])
def test_vmult_physical_types_generic(phys):
    case = Case((3, 3, 3), k=2, physical_type=phys, viscosity=0.1)
    eu, ep = run_vmult(case, variant=0)
    assert eu < TOL and ep < TOL, (eu, ep)


def test_vmult_variable_coefficients_generic():
    case = Case((4, 4, 3), k=2, density_diff=0.5, damping=0.2)
    eu, ep = run_vmult(case, variant=0, coefficients=True)
    assert eu < TOL and ep < TOL, (eu, ep)


def test_vmult_partial_constraints_generic():
    # Dirichlet only on x-faces, pressure constrained on one face, no mean fix
    case = Case((4, 3, 3), k=2, faces_u=[0, 1], faces_p=[3], pressure_average_fix=False)
    eu, ep = run_vmult(case, variant=0)
    assert eu < TOL and ep < TOL, (eu, ep)


@pytest.mark.parametrize("lin", [0, 1, 2, 3])
def test_residual_and_linearization_state(lin):
    case = Case((3, 3, 4), k=2, linearization=lin, steps=3)
    src_u, src_p = case.smooth_u(0.1) + 0.01 * case.random_u(), case.smooth_p(0.1)
    old_u, oldold_u = case.smooth_u(0.05), case.smooth_u(0.0)
    lin_ref = np.zeros(case.n_cells * case.nq * 12)
    ref_u, ref_p = orc.ns_residual(case.mesh, case.k, case.prm, src_u, src_p, old_u, oldold_u,
                                   con_u=case.con_u, con_p=case.con_p, lin=lin_ref)
    op = case.engine()
    src = op.block_vector(src_u, src_p)
    rhs = op.block_vector()
    old = op.block_vector(old_u)
    oldold = op.block_vector(oldold_u)
    op.residual(rhs, src, None, old, oldold)
    got_u, got_p = rhs.numpy()
    assert rel_l2(got_u, ref_u) < TOL and rel_l2(got_p, ref_p) < TOL
    if lin != 3:
        got_lin = op.get_linearization().reshape(-1, 12)
        ref = lin_ref.reshape(-1, 12)
        ncomp = 12 if lin == 0 else 4  # Picard/semi-implicit store (u, div) only
        assert rel_l2(got_lin[:, :ncomp], ref[:, :ncomp]) < TOL


def test_velocity_vmult_and_fixed_point():
    case = Case((3, 4, 3), k=2)
    src_u = case.random_u()
    lin = case.random_lin()
    ref = orc.ns_velocity_vmult(case.mesh, case.k, case.prm, src_u, case.con_u, lin=lin)
    op = case.engine()
    op.set_linearization(lin)
    op.fix_linearization_point()
    op.set_linearization(case.random_lin())  # later state must not affect velocity_vmult
    src, dst = op.initialize_u_vector(src_u), op.initialize_u_vector()
    op.velocity_vmult(dst, src)
    assert rel_l2(dst.numpy(), ref) < TOL


@pytest.mark.parametrize("k,ncell,variant", [(2, (3, 2, 4), 0), (3, (3, 2, 4), 0), (2, (3, 2, 4), 1),
                                             (2, (18, 17, 35), 1), (6, (2, 2, 2), 0)])
def test_scalar_sub_blocks(k, ncell, variant):
    """variant 1 with k = 2: constant-coefficient pressure mass / Poisson run on the structured
    Q1 sweep kernel, everything else on the generic kernels"""
    case = Case(ncell, k=k, faces_p=[0], density_diff=0.3, upper=(1.0, 0.7, 1.5))
    op = case.engine()
    op.set_kernel_variant(variant)
    src_u, src_p = case.random_u(), case.random_p()
    rho, mu, damp = case.random_coefficients()
    su, sp = op.initialize_u_vector(src_u), op.initialize_p_vector(src_p)
    for variable in (False, True):
        if variable:
            op.set_coefficients(rho, mu, damp)
        c = dict(rho=rho, mu=mu) if variable else dict(rho=None, mu=None)
        # divergence_vmult_add keeps the previous content of dst
        base = case.random_p()
        for wv in (False, True):
            dp = op.initialize_p_vector(base)
            op.divergence_vmult_add(dp, su, wv)
            ref = orc.ns_divergence_vmult_add(case.mesh, k, case.prm, src_u, base, case.con_u,
                                              case.con_p, mu=c["mu"], weight_by_viscosity=wv)
            assert rel_l2(dp.numpy(), ref) < TOL
        dp = op.initialize_p_vector(base)
        op.pressure_poisson_vmult(dp, sp)
        assert rel_l2(dp.numpy(), orc.ns_pressure_poisson_vmult(case.mesh, k, case.prm, src_p,
                                                                 case.con_p, rho=c["rho"])) < TOL
        op.pressure_mass_vmult(dp, sp)
        assert rel_l2(dp.numpy(), orc.ns_pressure_mass_vmult(case.mesh, k, case.prm, src_p,
                                                              case.con_p, mu=c["mu"])) < TOL
        op.set_linearization(case.random_lin())
        op.pressure_convdiff_vmult(dp, sp)
        assert rel_l2(dp.numpy(), orc.ns_pressure_convdiff_vmult(case.mesh, k, case.prm, src_p,
                                                                  case.con_p, mu=c["mu"])) < TOL


@pytest.mark.parametrize("ncell,faces_u,faces_p,lin", [((3, 2, 4), range(6), (), 0), ((40, 10, 5), (0, 3, 4), (1,), 0),
                                                        ((33, 9, 20), (), (2, 5), 0), ((34, 17, 9), (1, 2, 5), (), 0),
                                                        ((5, 4, 3), range(6), (), 4), ((1, 1, 1), (), (), 0),
                                                        ((1, 2, 1), (2,), (3,), 0), ((65, 1, 17), (0, 1), (), 0),
                                                        # pressure rows of exactly 62 / 124 nodes (the owned lanes of a wave)
                                                        ((61, 3, 2), (0, 1, 2), (4,), 0), ((123, 2, 13), (3, 4, 5), (0,), 0)])
@pytest.mark.parametrize("variant", [1, 2])
def test_divergence_as_tensor_product_stencil(ncell, faces_u, faces_p, lin, variant):
    """variant 1: Q2 -> Q1 stencil kernel (csrc/ns_divergence.hip), variant 2: divergence mode of the sweep
    kernel; partial Dirichlet sets, tiles cut by the mesh, the plain read of the projection scheme"""
    case = Case(ncell, k=2, faces_u=faces_u, faces_p=faces_p, upper=(1.0, 0.7, 1.5), viscosity=0.37, linearization=lin)
    op = case.engine()
    op.set_kernel_variant(variant)
    src_u, base = case.random_u(), case.random_p()
    su = op.initialize_u_vector(src_u)
    for wv in (False, True):
        dp = op.initialize_p_vector(base)
        op.divergence_vmult_add(dp, su, wv)
        ref = orc.ns_divergence_vmult_add(case.mesh, 2, case.prm, src_u, base, case.con_u, case.con_p, mu=None,
                                          weight_by_viscosity=wv)
        assert rel_l2(dp.numpy(), ref) < TOL


def test_beltrami_golden_residual_on_device(oracle):
    """tests/beltrami_3d.output:13 reproduced by the HIP residual kernel."""
    case = Case((16, 16, 16), k=2, steps=1)
    xu = orc.node_coordinates(case.mesh, 2)
    u0, p0 = case.smooth_u(0.0), case.smooth_p(0.0)
    sol_u = u0.copy()
    ub = orc.beltrami_u(xu, 0.05)
    sol_u[case.con_u == 1] = ub[case.con_u == 1]
    op = case.engine()
    rhs = op.block_vector()
    op.residual(rhs, op.block_vector(sol_u, p0), None, op.block_vector(u0), op.block_vector())
    op.apply_pressure_average_projection(rhs.block(1))
    ru, rp = rhs.numpy()
    assert abs(np.linalg.norm(ru) - 2.590) < 5e-4
    assert abs(np.linalg.norm(rp) - 6.423e-2) < 5e-6


def test_matvec_statistics_and_errors():
    case = Case((2, 2, 2), k=2)
    op = case.engine()
    src, dst = op.block_vector(case.random_u(), case.random_p()), op.block_vector()
    with pytest.raises(Exception):
        op.vmult(dst, src)  # linearization not set -> ExcNotInitialized-like error
    op.set_linearization(case.random_lin())
    op.get_matvec_statistics()
    for _ in range(3):
        op.vmult(dst, src)
    seconds, count = op.get_matvec_statistics()
    assert count == 3 and seconds > 0
    assert op.get_matvec_statistics()[1] == 0


# ----------------------------------------------------------------------------- Q2/Q1 sweep kernel
@pytest.mark.parametrize("ncell", [(8, 8, 4), (5, 3, 2), (16, 8, 9), (9, 17, 5), (24, 16, 20), (16, 10, 6), (10, 16, 7), (17, 9, 33),
                                   (1, 1, 1), (1, 2, 1), (2, 1, 3), (1, 1, 40)])
def test_vmult_q2_kernel_newton(ncell):
    case = Case(ncell, k=2, upper=(1.0, 0.5, 2.0))
    eu, ep = run_vmult(case, variant=1)
    assert eu < TOL and ep < TOL, (eu, ep)


@pytest.mark.parametrize("lin", [1, 2, 3, 4])
def test_vmult_q2_kernel_linearizations(lin):
    case = Case((9, 8, 5), k=2, linearization=lin, beta=1.0, tau_grad_div=0.3, damping=0.1)
    eu, ep = run_vmult(case, variant=1)
    assert eu < TOL and ep < TOL, (eu, ep)


@pytest.mark.parametrize("phys", [1, 2])
def test_vmult_q2_kernel_physical_types(phys):
    case = Case((8, 9, 3), k=2, physical_type=phys, viscosity=0.1)
    eu, ep = run_vmult(case, variant=1)
    assert eu < TOL and ep < TOL, (eu, ep)


def test_vmult_q2_kernel_partial_constraints():
    case = Case((10, 8, 6), k=2, faces_u=[0, 3, 4], faces_p=[1, 5], pressure_average_fix=False)
    eu, ep = run_vmult(case, variant=1)
    assert eu < TOL and ep < TOL, (eu, ep)


def test_velocity_vmult_q2_kernel():
    case = Case((9, 8, 7), k=2)
    src_u = case.random_u()
    lin = case.random_lin()
    ref = orc.ns_velocity_vmult(case.mesh, case.k, case.prm, src_u, case.con_u, lin=lin)
    op = case.engine()
    op.set_kernel_variant(1)
    op.set_linearization(lin)
    op.fix_linearization_point()
    op.set_linearization(case.random_lin())
    src, dst = op.initialize_u_vector(src_u), op.initialize_u_vector(np.full(case.n_u, 3.0))
    op.velocity_vmult(dst, src)
    assert rel_l2(dst.numpy(), ref) < TOL


def test_halo_pack_unpack_kernel_matches_slicing():
    """adaflo_halo_transfer (pack / unpack-copy / unpack-add of interface node boxes) vs numpy slicing"""
    import ctypes as C
    from adaflo_amd import _lib, parallel
    case = Case((3, 2, 4), k=2)
    op = case.engine()
    part = parallel.BrickPartition((3, 3, 3), 13, [3, 2, 4], [-1] * 3, [1] * 3)   # interior rank: 26 neighbours
    nn = part.nodes(2)
    v = case.random_u()
    offs = [o for o, _ in part.neighbours()]
    assert len(offs) == 26
    regs, ref = [], []
    v4 = v.reshape(nn[2], nn[1], nn[0], 3)
    for o in offs:
        sl = parallel._region(o, nn)
        ref.append(v4[sl].reshape(-1))
        for d in range(3):
            lo = 0 if o[d] <= 0 else nn[d] - 1
            regs += [lo, nn[d] if o[d] == 0 else lo + 1]
    ref = np.concatenate(ref)
    dv = op.initialize_u_vector(v)
    from adaflo_amd.vectors import DeviceVector
    buf = DeviceVector(op._ctx, ref.size)
    arr, nn_c = (C.c_int * len(regs))(*regs), (C.c_int * 3)(*nn)
    lib = _lib.load()
    _lib.check(op._ctx, lib.adaflo_halo_transfer(op._ctx, dv.ptr, buf.ptr, nn_c, 3, 26, arr, 0))
    assert np.array_equal(buf.numpy(), ref)
    # unpack-add into zeros: every interface node receives the sum over the regions containing it
    dz = op.initialize_u_vector()
    _lib.check(op._ctx, lib.adaflo_halo_transfer(op._ctx, dz.ptr, buf.ptr, nn_c, 3, 26, arr, 2))
    expect = np.zeros_like(v4)
    for o in offs:
        expect[parallel._region(o, nn)] += v4[parallel._region(o, nn)]
    assert rel_l2(dz.numpy(), expect.reshape(-1)) < 1e-15
    # unpack-copy of the six faces only
    faces = [i for i, o in enumerate(offs) if sum(abs(x) for x in o) == 1]
    farr = (C.c_int * (6 * len(faces)))(*[regs[6 * i + j] for i in faces for j in range(6)])
    fbuf = DeviceVector.from_numpy(op._ctx, np.concatenate([v4[parallel._region(offs[i], nn)].reshape(-1) for i in faces]))
    dz = op.initialize_u_vector()
    _lib.check(op._ctx, lib.adaflo_halo_transfer(op._ctx, dz.ptr, fbuf.ptr, nn_c, 3, len(faces), farr, 1))
    expect = np.zeros_like(v4)
    for i in faces:
        expect[parallel._region(offs[i], nn)] = v4[parallel._region(offs[i], nn)]
    assert np.array_equal(dz.numpy(), expect.reshape(-1))


def test_vmult_q2_kernel_random_bricks():
    """many brick shapes (partial tiles in x / y, short and ragged z-chunks, random Dirichlet faces)"""
    rng = np.random.default_rng(2026)
    for trial in range(14):
        ncell = tuple(int(x) for x in rng.integers(1, 21, 3))
        faces_u = [f for f in range(6) if rng.random() < 0.6]
        faces_p = [f for f in range(6) if rng.random() < 0.2]
        case = Case(ncell, k=2, upper=(1.0, 0.7, 1.3), faces_u=faces_u, faces_p=faces_p,
                    pressure_average_fix=bool(trial % 2), seed=trial)
        eu, ep = run_vmult(case, variant=1)
        assert eu < TOL and ep < TOL, (ncell, faces_u, faces_p, eu, ep)


# ----------------------------------------------------------------------------- BASELINE size
def test_full_size_properties_128cubed():
    """Config 2 (128^3 Q2/Q1, 53 M DoF) is far beyond what the naive oracle finishes in seconds,
    so parity at that size is checked through size-independent properties:
      * the Q2/Q1 sweep kernel and the generic per-cell kernel (independent code) agree,
      * linearity  A(a x + b y) = a A x + b A y,
      * constant pressure: B^T 1 vanishes on unconstrained velocity rows,
      * the Stokes operator (no projection) is symmetric: x^T A y = y^T A x.
    Tolerance 1e-12 relative L2 as everywhere."""
    n = 128
    case = Case((n, n, n), k=2)
    op = case.engine()
    rng = np.random.default_rng(7)
    # linearisation state = Beltrami interpolant seen through the residual kernel (as bench.py)
    u0 = case.smooth_u(0.0)
    tmp = op.block_vector()
    op.residual(tmp, op.block_vector(u0, case.smooth_p(0.0)), None, op.block_vector(u0), op.block_vector())
    x_u, x_p = rng.uniform(-1, 1, case.n_u), rng.uniform(-1, 1, case.n_p)
    y_u, y_p = rng.uniform(-1, 1, case.n_u), rng.uniform(-1, 1, case.n_p)
    x, y = op.block_vector(x_u, x_p), op.block_vector(y_u, y_p)
    dst = op.block_vector()
    res = {}
    for variant in (1, 0):
        op.set_kernel_variant(variant)
        op.vmult(dst, x)
        res[variant] = dst.numpy()
    assert rel_l2(res[1][0], res[0][0]) < TOL and rel_l2(res[1][1], res[0][1]) < TOL
    op.set_kernel_variant(1)
    ax_u, ax_p = res[1]
    op.vmult(dst, y)
    ay_u, ay_p = dst.numpy()
    a, b = 0.75, -1.25
    op.vmult(dst, op.block_vector(a * x_u + b * y_u, a * x_p + b * y_p))
    z_u, z_p = dst.numpy()
    assert rel_l2(z_u, a * ax_u + b * ay_u) < TOL and rel_l2(z_p, a * ax_p + b * ay_p) < TOL
    # constant pressure, zero velocity: only constrained rows (identity on a zero vector) and
    # round-off remain in the velocity block
    op.vmult(dst, op.block_vector(np.zeros(case.n_u), np.ones(case.n_p)))
    bt_u, _ = dst.numpy()
    h = 2.0 / n
    assert np.abs(bt_u).max() < 1e-12 * h * h
    del op
    # Stokes, no mean projection: symmetric operator (constrained rows are +-identity)
    sc = Case((n, n, n), k=2, physical_type=2, pressure_average_fix=False)
    sop = sc.engine()
    d1, d2 = sop.block_vector(), sop.block_vector()
    sop.vmult(d1, sop.block_vector(x_u, x_p))
    sop.vmult(d2, sop.block_vector(y_u, y_p))
    a1u, a1p = d1.numpy()
    a2u, a2p = d2.numpy()
    xay = y_u @ a1u + y_p @ a1p
    yax = x_u @ a2u + x_p @ a2p
    assert abs(xay - yax) < 1e-12 * max(abs(xay), np.linalg.norm(a1u) * np.linalg.norm(y_u))


# ----------------------------------------------------------------------------- two-phase flow
@pytest.mark.parametrize("lin,ncell", [(0, (4, 4, 3)), (0, (17, 9, 20)), (1, (9, 16, 5)), (2, (10, 8, 6)), (3, (9, 8, 5)), (3, (17, 9, 20))])
def test_vmult_variable_coefficients_q2_kernel(lin, ncell):
    """variable rho / mu / damping at the quadrature points ride in the spare lanes of the state
    pieces of the Q2/Q1 sweep kernel (navier_stokes_matrix.cc:636-642,:827-845); round 6: the explicit scheme (3) has no
    state for them to ride on -- the kernel reads them from the generic arrays, one array per lane of a quad"""
    case = Case(ncell, k=2, linearization=lin, density_diff=0.5, damping=0.2, tau_grad_div=0.1,
                upper=(1.0, 0.5, 2.0))
    eu, ep = run_vmult(case, variant=1, coefficients=True)
    assert eu < TOL and ep < TOL, (eu, ep)


def test_velocity_vmult_variable_coefficients_uses_the_frozen_state():
    case = Case((9, 8, 7), k=2, density_diff=0.5)
    src_u, lin, coef = case.random_u(), case.random_lin(), case.random_coefficients()
    ref = orc.ns_velocity_vmult(case.mesh, case.k, case.prm, src_u, case.con_u, lin=lin,
                                rho=coef[0], mu=coef[1], damp=coef[2])
    for variant in (1, 0):
        op = case.engine()
        op.set_kernel_variant(variant)
        op.set_linearization(lin)
        op.set_coefficients(*coef)
        op.fix_linearization_point()
        # later changes of the state and the coefficients must not affect velocity_vmult (:349-375)
        op.set_linearization(case.random_lin())
        op.set_coefficients(*case.random_coefficients())
        src, dst = op.initialize_u_vector(src_u), op.initialize_u_vector(np.full(case.n_u, 3.0))
        op.velocity_vmult(dst, src)
        assert rel_l2(dst.numpy(), ref) < TOL
        # ... while vmult sees the new ones: switching back to constant coefficients works too
        op.set_coefficients(None, None, None)


def test_explicit_scheme_with_variable_coefficients_on_the_sweep_kernel():
    """two-phase flow with explicit convection (rising_bubble_ls_expl.prm of the reference): vmult and velocity_vmult have no
    linearisation state; velocity_vmult takes the coefficients fix_linearization_point froze while vmult sees the new ones,
    constant coefficients again after they are cleared; sweep kernel (1) and generic kernel (0) against the oracle"""
    case = Case((9, 8, 5), k=2, linearization=3, density_diff=0.5, damping=0.2, tau_grad_div=0.1, upper=(1.0, 0.5, 2.0), steps=3)
    src_u, src_p = case.random_u(), case.random_p()
    co1, co2 = case.random_coefficients(), case.random_coefficients()
    w, modes = case.weights_modes()
    ref_vel1 = orc.ns_velocity_vmult(case.mesh, 2, case.prm, src_u, case.con_u, rho=co1[0], mu=co1[1], damp=co1[2])
    ref2 = orc.ns_vmult(case.mesh, 2, case.prm, src_u, src_p, case.con_u, case.con_p, rho=co2[0], mu=co2[1], damp=co2[2],
                        weights=w, modes=modes)
    ref3 = orc.ns_vmult(case.mesh, 2, case.prm, src_u, src_p, case.con_u, case.con_p, weights=w, modes=modes)
    for variant in (1, 0):
        op = case.engine()
        op.set_kernel_variant(variant)
        op.set_coefficients(*co1)
        op.fix_linearization_point()
        op.set_coefficients(*co2)
        vsrc, vdst = op.initialize_u_vector(src_u), op.initialize_u_vector(np.full(case.n_u, 3.0))
        op.velocity_vmult(vdst, vsrc)
        assert rel_l2(vdst.numpy(), ref_vel1) < TOL, variant
        dst = op.block_vector()
        op.vmult(dst, op.block_vector(src_u, src_p))
        gu, gp = dst.numpy()
        assert rel_l2(gu, ref2[0]) < TOL and rel_l2(gp, ref2[1]) < TOL, variant
        op.set_coefficients(None, None, None)
        op.vmult(dst, op.block_vector(src_u, src_p))
        gu, gp = dst.numpy()
        assert rel_l2(gu, ref3[0]) < TOL and rel_l2(gp, ref3[1]) < TOL, variant
        op.velocity_vmult(vdst, vsrc)                          # (still the frozen two-phase operator)
        assert rel_l2(vdst.numpy(), ref_vel1) < TOL, variant


@pytest.mark.parametrize("k,ncell,lin,phys,coefficients,faces_u",
                         [(2, (3, 2, 2), 0, 0, True, range(6)), (2, (2, 3, 2), 1, 0, True, (0, 3)), (2, (2, 2, 2), 4, 0, False, ()),
                          (2, (2, 2, 3), 0, 2, True, (1, 4)), (3, (2, 2, 1), 0, 0, True, range(6)), (3, (2, 1, 2), 0, 1, False, (2,)), (3, (1, 2, 2), 2, 0, True, ()),
                          (4, (1, 2, 1), 0, 0, True, (5,))])
def test_velocity_block_diagonal_is_the_diagonal_of_velocity_vmult(k, ncell, lin, phys, coefficients, faces_u):
    """adaflo_ns_velocity_block_diagonal (cell-wise from the quadrature-point operation) against column by
    column applications of velocity_vmult to unit vectors, for every linearisation branch of the kernel,
    frozen variable coefficients and partial Dirichlet sets"""
    case = Case(ncell, k=k, linearization=lin, physical_type=phys, faces_u=faces_u, density_diff=0.5 if coefficients else 0.0,
                damping=0.3, tau_grad_div=0.2, upper=(1.0, 0.6, 1.7))
    op = case.engine()
    op.set_kernel_variant(0)
    op.set_linearization(case.random_lin())
    if coefficients:
        op.set_coefficients(*case.random_coefficients())
    diag = op.initialize_u_vector(np.full(case.n_u, 7.0))
    op.velocity_block_diagonal(diag)
    got = diag.numpy()
    ref = np.empty(case.n_u)
    e = np.zeros(case.n_u)
    dst = op.initialize_u_vector(e)
    for i in range(case.n_u):
        e[:] = 0.0
        e[i] = 1.0
        op.velocity_vmult(dst, op.initialize_u_vector(e))
        ref[i] = dst.numpy()[i]
    assert np.max(np.abs(got - ref)) < 1e-12 * np.max(np.abs(ref)), np.max(np.abs(got - ref))


# ----------------------------------------------------------------------------- Q3..Q5 sweep kernel
@pytest.mark.parametrize("k,ncell", [(3, (4, 4, 3)), (3, (9, 5, 6)), (3, (8, 8, 20)), (4, (4, 2, 3)), (4, (5, 3, 4)),
                                     (4, (9, 6, 10)), (5, (3, 2, 2)), (5, (4, 3, 3))])
def test_vmult_high_order_sweep_kernel(k, ncell):
    """tiles, partial tiles, several z-chunks; Newton with pressure"""
    case = Case(ncell, k=k, upper=(1.0, 0.5, 2.0), tau_grad_div=0.2)
    eu, ep = run_vmult(case, variant=2)
    assert eu < TOL and ep < TOL, (eu, ep)


@pytest.mark.parametrize("lin,phys", [(1, 0), (2, 0), (3, 0), (4, 0), (0, 1), (0, 2)])
def test_vmult_high_order_sweep_kernel_modes(lin, phys):
    case = Case((5, 4, 3), k=3, linearization=lin, physical_type=phys, beta=1.0, tau_grad_div=0.3, viscosity=0.2)
    eu, ep = run_vmult(case, variant=2)
    assert eu < TOL and ep < TOL, (eu, ep)


def test_vmult_high_order_sweep_kernel_partial_constraints():
    case = Case((5, 5, 3), k=4, faces_u=[0, 3, 4], faces_p=[1, 5], pressure_average_fix=False)
    eu, ep = run_vmult(case, variant=2)
    assert eu < TOL and ep < TOL, (eu, ep)


def test_velocity_vmult_high_order_sweep_kernel():
    case = Case((5, 3, 4), k=4)
    src_u, lin = case.random_u(), case.random_lin()
    ref = orc.ns_velocity_vmult(case.mesh, case.k, case.prm, src_u, case.con_u, lin=lin)
    op = case.engine()
    op.set_kernel_variant(2)
    op.set_linearization(lin)
    op.fix_linearization_point()
    op.set_linearization(case.random_lin())
    src, dst = op.initialize_u_vector(src_u), op.initialize_u_vector(np.full(case.n_u, 3.0))
    op.velocity_vmult(dst, src)
    assert rel_l2(dst.numpy(), ref) < TOL


# ----------------------------------------------------------------------------- Q4/Q3 plane-per-lane kernel (round 5)
@pytest.mark.parametrize("ncell", [(4, 2, 3), (5, 3, 4), (9, 6, 10), (33, 5, 9), (1, 1, 1), (2, 7, 2)])
def test_vmult_plane_per_lane_kernel(ncell):
    """csrc/ns_hop.hip (variant 3, k = 4): full and clipped 2 x 2 tiles, Newton with pressure"""
    case = Case(ncell, k=4, upper=(1.0, 0.5, 2.0), tau_grad_div=0.2)
    eu, ep = run_vmult(case, variant=3)
    assert eu < TOL and ep < TOL, (eu, ep)


@pytest.mark.parametrize("lx", [1, 2, 3, 7])
def test_vmult_plane_per_lane_kernel_chunks(lx):
    """x-chunks of every length: the seam planes between chunks go through the x-slabs"""
    case = Case((7, 5, 9), k=4, tau_grad_div=0.1)
    src_u, src_p, lin = case.random_u(), case.random_p(), case.random_lin()
    w, modes = case.weights_modes()
    ref_u, ref_p = orc.ns_vmult(case.mesh, case.k, case.prm, src_u, src_p, case.con_u, case.con_p, lin=lin,
                                weights=w, modes=modes)
    op = case.engine()
    op.set_kernel_variant(3)
    op.set_linearization(lin)
    op.set_x_chunk(lx)
    dst = op.block_vector(np.full(case.n_u, 7.0), np.full(case.n_p, 7.0))
    op.vmult(dst, op.block_vector(src_u, src_p))
    got_u, got_p = dst.numpy()
    assert rel_l2(got_u, ref_u) < TOL and rel_l2(got_p, ref_p) < TOL


@pytest.mark.parametrize("lin,phys", [(1, 0), (2, 0), (3, 0), (4, 0), (0, 1), (0, 2)])
def test_vmult_plane_per_lane_kernel_modes(lin, phys):
    case = Case((5, 4, 3), k=4, linearization=lin, physical_type=phys, beta=1.0, tau_grad_div=0.3, viscosity=0.2,
                damping=0.3)
    eu, ep = run_vmult(case, variant=3)
    assert eu < TOL and ep < TOL, (eu, ep)


def test_vmult_plane_per_lane_kernel_partial_constraints():
    case = Case((5, 5, 3), k=4, faces_u=[0, 3, 4], faces_p=[1, 5], pressure_average_fix=False)
    eu, ep = run_vmult(case, variant=3)
    assert eu < TOL and ep < TOL, (eu, ep)


def test_velocity_vmult_plane_per_lane_kernel():
    """velocity_vmult on the state frozen by fix_linearization_point; the streaming copies of the state follow the
    generic ones; variable coefficients fall back to the x-marching kernel under the same variant"""
    case = Case((5, 3, 4), k=4)
    src_u, src_p, lin, lin2 = case.random_u(), case.random_p(), case.random_lin(), case.random_lin()
    ref = orc.ns_velocity_vmult(case.mesh, case.k, case.prm, src_u, case.con_u, lin=lin)
    w, modes = case.weights_modes()
    ref2_u, ref2_p = orc.ns_vmult(case.mesh, case.k, case.prm, src_u, src_p, case.con_u, case.con_p, lin=lin2,
                                  weights=w, modes=modes)
    op = case.engine()
    op.set_kernel_variant(3)
    op.set_linearization(lin)
    dst2 = op.block_vector()
    op.vmult(dst2, op.block_vector(src_u, src_p))              # (builds the streaming copy of `lin`)
    op.fix_linearization_point()
    op.set_linearization(lin2)
    src, dst = op.initialize_u_vector(src_u), op.initialize_u_vector(np.full(case.n_u, 3.0))
    op.velocity_vmult(dst, src)
    assert rel_l2(dst.numpy(), ref) < TOL
    op.vmult(dst2, op.block_vector(src_u, src_p))
    got_u, got_p = dst2.numpy()
    assert rel_l2(got_u, ref2_u) < TOL and rel_l2(got_p, ref2_p) < TOL
    eu, ep = run_vmult(Case((3, 4, 3), k=4, density_diff=0.5), variant=3, coefficients=True)
    assert eu < TOL and ep < TOL, (eu, ep)


# ----------------------------------------------------------------------------- Q3..Q5 x-marching kernel (round 4)
@pytest.mark.parametrize("k,ncell", [(3, (4, 4, 3)), (3, (9, 5, 6)), (3, (8, 8, 20)), (4, (4, 2, 3)), (4, (5, 3, 4)),
                                     (4, (9, 6, 10)), (4, (33, 5, 9)), (5, (3, 2, 2)), (5, (4, 3, 3))])
def test_vmult_high_order_x_marching_kernel(k, ncell):
    """csrc/ns_hox.hip (variant 1 for k >= 3): cross-sections, partial cross-sections, Newton with pressure"""
    case = Case(ncell, k=k, upper=(1.0, 0.5, 2.0), tau_grad_div=0.2)
    eu, ep = run_vmult(case, variant=1)
    assert eu < TOL and ep < TOL, (eu, ep)


@pytest.mark.parametrize("lx", [1, 2, 3, 7])
def test_vmult_high_order_x_marching_kernel_chunks(lx):
    """x-chunks of every length: the seam planes between chunks go through the x-slabs"""
    case = Case((7, 5, 9), k=4, tau_grad_div=0.1)
    src_u, src_p, lin = case.random_u(), case.random_p(), case.random_lin()
    w, modes = case.weights_modes()
    ref_u, ref_p = orc.ns_vmult(case.mesh, case.k, case.prm, src_u, src_p, case.con_u, case.con_p, lin=lin,
                                weights=w, modes=modes)
    op = case.engine()
    op.set_linearization(lin)
    op.set_x_chunk(lx)
    dst = op.block_vector(np.full(case.n_u, 7.0), np.full(case.n_p, 7.0))
    op.vmult(dst, op.block_vector(src_u, src_p))
    got_u, got_p = dst.numpy()
    assert rel_l2(got_u, ref_u) < TOL and rel_l2(got_p, ref_p) < TOL


@pytest.mark.parametrize("k", [3, 4, 5])
@pytest.mark.parametrize("lin,phys", [(1, 0), (2, 0), (3, 0), (4, 0), (0, 1), (0, 2)])
def test_vmult_high_order_x_marching_kernel_modes(k, lin, phys):
    case = Case((5, 4, 3), k=k, linearization=lin, physical_type=phys, beta=1.0, tau_grad_div=0.3, viscosity=0.2,
                damping=0.3)
    eu, ep = run_vmult(case, variant=1)
    assert eu < TOL and ep < TOL, (eu, ep)


@pytest.mark.parametrize("k", [3, 4, 5])
def test_vmult_high_order_x_marching_kernel_partial_constraints(k):
    case = Case((5, 5, 3), k=k, faces_u=[0, 3, 4], faces_p=[1, 5], pressure_average_fix=False)
    eu, ep = run_vmult(case, variant=1)
    assert eu < TOL and ep < TOL, (eu, ep)


def test_velocity_vmult_high_order_x_marching_kernel():
    """velocity_vmult on the state frozen by fix_linearization_point; the streaming copies of the state follow the
    generic ones (a new set_linearization must not leak into the frozen operator, and must reach vmult)"""
    case = Case((5, 3, 4), k=4)
    src_u, src_p, lin, lin2 = case.random_u(), case.random_p(), case.random_lin(), case.random_lin()
    ref = orc.ns_velocity_vmult(case.mesh, case.k, case.prm, src_u, case.con_u, lin=lin)
    op = case.engine()
    op.set_kernel_variant(1)
    op.set_linearization(lin)
    dst2 = op.block_vector()
    op.vmult(dst2, op.block_vector(src_u, src_p))              # (builds the streaming copy of `lin`)
    op.fix_linearization_point()
    op.set_linearization(lin2)
    src, dst = op.initialize_u_vector(src_u), op.initialize_u_vector(np.full(case.n_u, 3.0))
    op.velocity_vmult(dst, src)
    assert rel_l2(dst.numpy(), ref) < TOL
    w, modes = case.weights_modes()
    ref_u, ref_p = orc.ns_vmult(case.mesh, case.k, case.prm, src_u, src_p, case.con_u, case.con_p, lin=lin2,
                                weights=w, modes=modes)
    op.vmult(dst2, op.block_vector(src_u, src_p))
    got_u, got_p = dst2.numpy()
    assert rel_l2(got_u, ref_u) < TOL and rel_l2(got_p, ref_p) < TOL


@pytest.mark.parametrize("k,ncell,lin,phys,chunk", [(4, (5, 4, 9), 0, 0, 0), (3, (6, 5, 5), 1, 0, 2), (5, (3, 2, 3), 0, 0, 0),
                                                    (4, (3, 5, 2), 0, 2, 1), (3, (9, 4, 3), 0, 0, 4), (4, (4, 3, 3), 2, 0, 0)])
def test_two_phase_vmult_x_marching_kernel(k, ncell, lin, phys, chunk):
    """variable density / viscosity / damping at the quadrature points on the Q3..Q5 x-marching kernel (template VARCO:
    the coefficients travel as two more pieces of the state stream): vmult and velocity block against the oracle, a
    frozen copy (state AND coefficients) that survives new coefficients, and the way back to constant coefficients"""
    case = Case(ncell, k=k, lower=(0., 0., 0.), upper=(1., 1.5, 1.), faces_u=[0, 2, 3, 5], faces_p=[1],
                linearization=lin, physical_type=phys, tau_grad_div=0.2, density_diff=-0.5, steps=3)
    src_u, src_p, lin_q = case.random_u(), case.random_p(), case.random_lin()
    rho, mu, damp = case.random_coefficients()
    rho2, mu2, damp2 = case.random_coefficients()
    w, modes = case.weights_modes()
    ref_u, ref_p = orc.ns_vmult(case.mesh, k, case.prm, src_u, src_p, case.con_u, case.con_p, lin=lin_q, rho=rho, mu=mu,
                                damp=damp, weights=w, modes=modes)
    ref_vel = orc.ns_velocity_vmult(case.mesh, k, case.prm, src_u, case.con_u, lin=lin_q, rho=rho, mu=mu, damp=damp)
    ref2_u, ref2_p = orc.ns_vmult(case.mesh, k, case.prm, src_u, src_p, case.con_u, case.con_p, lin=lin_q, rho=rho2, mu=mu2,
                                  damp=damp2, weights=w, modes=modes)
    ref3_u, ref3_p = orc.ns_vmult(case.mesh, k, case.prm, src_u, src_p, case.con_u, case.con_p, lin=lin_q, weights=w, modes=modes)
    for variant in (1, 0):
        op = case.engine()
        op.set_kernel_variant(variant)
        op.set_x_chunk(chunk)
        if phys != 2:
            op.set_linearization(lin_q)
        op.set_coefficients(rho, mu, damp)
        src, dst = op.block_vector(src_u, src_p), op.block_vector()
        op.vmult(dst, src)
        gu, gp = dst.numpy()
        assert rel_l2(gu, ref_u) < TOL and rel_l2(gp, ref_p) < TOL, (variant, rel_l2(gu, ref_u), rel_l2(gp, ref_p))
        op.fix_linearization_point()
        op.set_coefficients(rho2, mu2, damp2)
        vdst = op.initialize_u_vector()
        op.velocity_vmult(vdst, src.block(0))                   # frozen coefficients
        assert rel_l2(vdst.numpy(), ref_vel) < TOL, variant
        op.vmult(dst, src)                                      # current coefficients
        gu, gp = dst.numpy()
        assert rel_l2(gu, ref2_u) < TOL and rel_l2(gp, ref2_p) < TOL, variant
        op.set_coefficients(None, None, None)
        op.vmult(dst, src)
        gu, gp = dst.numpy()
        assert rel_l2(gu, ref3_u) < TOL and rel_l2(gp, ref3_p) < TOL, variant
        op.velocity_vmult(vdst, src.block(0))                   # (still the frozen two-phase operator)
        assert rel_l2(vdst.numpy(), ref_vel) < TOL, variant


@pytest.mark.parametrize("k,ncell,lin,phys,chunk", [(4, (5, 4, 9), 0, 0, 0), (4, (9, 3, 2), 1, 0, 4), (3, (6, 5, 5), 0, 0, 2),
                                                    (5, (3, 2, 3), 0, 0, 0), (4, (4, 4, 4), 0, 1, 0), (4, (3, 5, 2), 0, 2, 1),
                                                    (3, (4, 4, 3), 1, 0, 0), (4, (5, 4, 9), 2, 0, 0), (3, (6, 5, 5), 2, 0, 2),
                                                    (5, (3, 2, 3), 2, 0, 0), (4, (4, 4, 4), 3, 0, 1), (3, (3, 4, 5), 3, 0, 0),
                                                    (4, (5, 4, 9), 4, 0, 0), (3, (6, 5, 5), 4, 0, 2), (5, (3, 2, 3), 4, 0, 0)])
def test_residual_x_marching_kernel(k, ncell, lin, phys, chunk):
    """residual mode of the Q3..Q5 x-marching kernel (round 4; navier_stokes_matrix.cc:266-293, 663-686, 725-800): the
    right-hand side with the read-modify-write semantics of the reference, partial constraints whose boundary values are
    read plainly, the state it leaves in the STREAMING layout only -- read back through the generic one, used by the
    next vmult, frozen by fix_linearization_point while a later residual replaces it --, Picard-type state, stationary
    and Stokes equations; (round 5) the schemes that linearise about the extrapolated old velocity (:740-782; semi-implicit
    = 2 stores (u_ext, div u_ext), explicit = 3 stores nothing); (round 6) the projection scheme = 4: the semi-implicit
    residual without the pressure rows (:902-907); the generic kernel on the same inputs"""
    case = Case(ncell, k=k, lower=(0., 0., 0.), upper=(1., 1.5, 1.), faces_u=[0, 2, 3, 5], faces_p=[1],
                linearization=lin, physical_type=phys, tau_grad_div=0.2, damping=0.1, density=1.2, steps=3)
    src_u, src_p = case.smooth_u(0.1) + 0.01 * case.random_u(), case.smooth_p(0.1)
    old_u, oldold_u = case.smooth_u(0.05), case.smooth_u(0.0)
    rhs0_u, rhs0_p, usr_u, usr_p = case.random_u(), case.random_p(), case.random_u(), case.random_p()
    lin_ref = np.zeros(case.n_cells * case.nq * 12)
    ref_u, ref_p = orc.ns_residual(case.mesh, k, case.prm, src_u, src_p, old_u, oldold_u, con_u=case.con_u, con_p=case.con_p,
                                   lin=lin_ref, rhs_u=rhs0_u, rhs_p=rhs0_p, user_u=usr_u, user_p=usr_p)
    vm_u, vm_p = case.random_u(), case.random_p()
    w, modes = case.weights_modes()
    ref_vu, ref_vp = orc.ns_vmult(case.mesh, k, case.prm, vm_u, vm_p, case.con_u, case.con_p, lin=lin_ref,
                                  weights=w, modes=modes)
    for variant in (1, 0):
        op = case.engine()
        op.set_kernel_variant(variant)
        op.set_x_chunk(chunk)
        rhs = op.block_vector(rhs0_u, rhs0_p)
        op.residual(rhs, op.block_vector(src_u, src_p), op.block_vector(usr_u, usr_p), op.block_vector(old_u),
                    op.block_vector(oldold_u))
        got_u, got_p = rhs.numpy()
        assert rel_l2(got_u, ref_u) < TOL and rel_l2(got_p, ref_p) < TOL, (variant, rel_l2(got_u, ref_u), rel_l2(got_p, ref_p))
        dst = op.block_vector()
        op.vmult(dst, op.block_vector(vm_u, vm_p))      # on the state the residual wrote
        gu, gp = dst.numpy()
        assert rel_l2(gu, ref_vu) < TOL and rel_l2(gp, ref_vp) < TOL, (variant, rel_l2(gu, ref_vu))
        if phys != 2 and lin != 3:
            ncomp = 12 if lin == 0 else 4
            got_lin = op.get_linearization().reshape(-1, 12)
            assert rel_l2(got_lin[:, :ncomp], lin_ref.reshape(-1, 12)[:, :ncomp]) < TOL
            op.vmult(dst, op.block_vector(vm_u, vm_p))  # (the streaming copy is still the current one)
            gu, gp = dst.numpy()
            assert rel_l2(gu, ref_vu) < TOL and rel_l2(gp, ref_vp) < TOL
            # frozen state of the preconditioner: velocity_vmult keeps using it after a new residual
            op.fix_linearization_point()
            ref_vel = orc.ns_velocity_vmult(case.mesh, k, case.prm, vm_u, case.con_u, lin=lin_ref)
            op.residual(rhs, op.block_vector(0.5 * src_u, src_p), None, op.block_vector(old_u), op.block_vector(oldold_u))
            vsrc, vdst = op.initialize_u_vector(vm_u), op.initialize_u_vector()
            op.velocity_vmult(vdst, vsrc)
            assert rel_l2(vdst.numpy(), ref_vel) < TOL, variant
            # ... and vmult uses the NEW state
            lin2 = np.zeros_like(lin_ref)
            orc.ns_residual(case.mesh, k, case.prm, 0.5 * src_u, src_p, old_u, oldold_u, con_u=case.con_u, con_p=case.con_p,
                            lin=lin2)
            r2u, r2p = orc.ns_vmult(case.mesh, k, case.prm, vm_u, vm_p, case.con_u, case.con_p, lin=lin2, weights=w, modes=modes)
            op.vmult(dst, op.block_vector(vm_u, vm_p))
            gu, gp = dst.numpy()
            assert rel_l2(gu, r2u) < TOL and rel_l2(gp, r2p) < TOL, variant
            op.set_kernel_variant(0)                      # generic kernel on the frozen streaming copy
            op.velocity_vmult(vdst, vsrc)
            assert rel_l2(vdst.numpy(), ref_vel) < TOL, variant
            op.vmult(dst, op.block_vector(vm_u, vm_p))
            gu, gp = dst.numpy()
            assert rel_l2(gu, r2u) < TOL and rel_l2(gp, r2p) < TOL, variant


@pytest.mark.parametrize("k", [3, 4, 5])
@pytest.mark.parametrize("lin", [0, 1, 2, 3])
def test_residual_x_marching_kernel_on_partial_tiles(k, lin):
    """the residual modes of the x-marching kernel (incl. the extrapolating schemes, which run with 512 registers per
    lane) on meshes whose workgroup cross-sections are cut in y, in z, in both, and on a single cell: waves without a
    cell of their own compute on cell (0, 0) and must leave no trace"""
    for ncell in ((1, 1, 1), (2, 2, 3), (3, 2, 5), (1, 3, 2), (2, 5, 1)):
        case = Case(ncell, k=k, lower=(0., 0., 0.), upper=(1., 1.5, 1.), faces_u=[0, 2, 3, 5], faces_p=[1],
                    linearization=lin, tau_grad_div=0.2, damping=0.1, density=1.2, steps=3)
        src_u, src_p = case.smooth_u(0.1) + 0.01 * case.random_u(), case.smooth_p(0.1)
        old_u, oldold_u = case.smooth_u(0.05), case.smooth_u(0.0)
        lin_ref = np.zeros(case.n_cells * case.nq * 12)
        ref_u, ref_p = orc.ns_residual(case.mesh, k, case.prm, src_u, src_p, old_u, oldold_u, con_u=case.con_u,
                                       con_p=case.con_p, lin=lin_ref)
        op = case.engine()
        for rep in range(2):
            rhs = op.block_vector()
            op.residual(rhs, op.block_vector(src_u, src_p), None, op.block_vector(old_u), op.block_vector(oldold_u))
            got_u, got_p = rhs.numpy()
            assert rel_l2(got_u, ref_u) < TOL and rel_l2(got_p, ref_p) < TOL, (ncell, rep, rel_l2(got_u, ref_u), rel_l2(got_p, ref_p))
        if lin != 3:
            ncomp = 12 if lin == 0 else 4
            got_lin = op.get_linearization().reshape(-1, 12)
            assert rel_l2(got_lin[:, :ncomp], lin_ref.reshape(-1, 12)[:, :ncomp]) < TOL, ncell


@pytest.mark.parametrize("k", [2, 4])
def test_change_of_scheme_after_the_state_was_frozen(k):
    """adaflo_ns_set_params with another linearisation while streaming copies of the state exist (the sweep kernels
    keep the state in a scheme-specific layout): vmult re-creates its copy for the new scheme, velocity_vmult keeps
    operating on the FROZEN state -- read the way the new scheme reads a stored state -- and a later
    fix_linearization_point freezes the current one again"""
    case = Case((9, 8, 3) if k == 2 else (4, 3, 3), k=k)
    picard = Case(case.ncell, k=k, linearization=1)
    src_u, src_p, lin, lin2 = case.random_u(), case.random_p(), case.random_lin(), case.random_lin()
    op = case.engine()
    op.set_kernel_variant(1)
    op.set_linearization(lin)
    dst2 = op.block_vector()
    op.vmult(dst2, op.block_vector(src_u, src_p))              # (streaming copy in the Newton layout)
    op.fix_linearization_point()
    op.set_linearization(lin2)
    op.vmult(dst2, op.block_vector(src_u, src_p))
    op.parameters = picard.fp
    op.update_parameters()
    src, dst = op.initialize_u_vector(src_u), op.initialize_u_vector(np.full(case.n_u, 3.0))
    op.velocity_vmult(dst, src)
    assert rel_l2(dst.numpy(), orc.ns_velocity_vmult(case.mesh, k, picard.prm, src_u, case.con_u, lin=lin)) < TOL
    w, modes = case.weights_modes()
    ref_u, ref_p = orc.ns_vmult(case.mesh, k, picard.prm, src_u, src_p, case.con_u, case.con_p, lin=lin2,
                                weights=w, modes=modes)
    op.vmult(dst2, op.block_vector(src_u, src_p))
    got_u, got_p = dst2.numpy()
    assert rel_l2(got_u, ref_u) < TOL and rel_l2(got_p, ref_p) < TOL
    op.fix_linearization_point()
    op.velocity_vmult(dst, src)
    assert rel_l2(dst.numpy(), orc.ns_velocity_vmult(case.mesh, k, picard.prm, src_u, case.con_u, lin=lin2)) < TOL


@pytest.mark.parametrize("ncell,upper,lin,phys", [((9, 8, 5), (1., 1., 1.), 0, 0), ((17, 9, 6), (1., 1., 3.), 0, 0),
                                                  ((8, 16, 3), (1., 1., 1.), 1, 0), ((5, 4, 9), (1., 2., 1.), 0, 1),
                                                  ((4, 5, 3), (1., 1., 1.), 0, 2), ((1, 1, 1), (1., 1., 1.), 0, 0),
                                                  ((9, 8, 5), (1., 1., 1.), 2, 0), ((17, 9, 6), (1., 1., 3.), 3, 0),
                                                  ((5, 4, 9), (1., 2., 1.), 2, 0), ((1, 1, 1), (1., 1., 1.), 2, 0),
                                                  ((9, 8, 5), (1., 1., 1.), 4, 0), ((17, 9, 6), (1., 1., 3.), 4, 0)])
def test_residual_sweep_kernel(ncell, upper, lin, phys):
    """residual mode of the Q2/Q1 sweep kernel (partial and multiple tiles, several z-chunks, non-cubic
    cells, Picard state, stationary and Stokes equations; round 5: the semi-implicit (2) and explicit (3) schemes, which
    linearise about the extrapolated old velocity, :740-782; round 6: the projection scheme (4), :902-907): right-hand side with the read-modify-write
    semantics of the reference (rhs = user - rhs - cell loop), the state it leaves (both layouts) and the
    operator applied on that state, against the oracle; generic kernel on the same inputs"""
    case = Case(ncell, k=2, lower=(0., 0., 0.), upper=upper, linearization=lin, physical_type=phys, steps=3)
    src_u, src_p = case.smooth_u(0.1) + 0.01 * case.random_u(), case.smooth_p(0.1)
    old_u, oldold_u = case.smooth_u(0.05), case.smooth_u(0.0)
    rhs0_u, rhs0_p, usr_u, usr_p = case.random_u(), case.random_p(), case.random_u(), case.random_p()
    lin_ref = np.zeros(case.n_cells * case.nq * 12)
    ref_u, ref_p = orc.ns_residual(case.mesh, case.k, case.prm, src_u, src_p, old_u, oldold_u,
                                   con_u=case.con_u, con_p=case.con_p, lin=lin_ref, rhs_u=rhs0_u, rhs_p=rhs0_p,
                                   user_u=usr_u, user_p=usr_p)
    vm_u, vm_p = case.random_u(), case.random_p()
    w, modes = case.weights_modes()
    ref_vu, ref_vp = orc.ns_vmult(case.mesh, 2, case.prm, vm_u, vm_p, case.con_u, case.con_p, lin=lin_ref,
                                  weights=w, modes=modes)
    for variant in (1, 0):
        op = case.engine()
        op.set_kernel_variant(variant)
        rhs = op.block_vector(rhs0_u, rhs0_p)
        op.residual(rhs, op.block_vector(src_u, src_p), op.block_vector(usr_u, usr_p), op.block_vector(old_u),
                    op.block_vector(oldold_u))
        got_u, got_p = rhs.numpy()
        assert rel_l2(got_u, ref_u) < TOL and rel_l2(got_p, ref_p) < TOL, (variant, rel_l2(got_u, ref_u))
        dst = op.block_vector()
        op.vmult(dst, op.block_vector(vm_u, vm_p))      # streaming state written by the residual
        gu, gp = dst.numpy()
        assert rel_l2(gu, ref_vu) < TOL and rel_l2(gp, ref_vp) < TOL, (variant, rel_l2(gu, ref_vu))
        if phys != 2 and lin != 3:
            ncomp = 12 if lin == 0 else 4
            got_lin = op.get_linearization().reshape(-1, 12)
            assert rel_l2(got_lin[:, :ncomp], lin_ref.reshape(-1, 12)[:, :ncomp]) < TOL
            # frozen state of the preconditioner: velocity_vmult keeps using it after a new residual
            op.fix_linearization_point()
            ref_vel = orc.ns_velocity_vmult(case.mesh, 2, case.prm, vm_u, case.con_u, lin=lin_ref)
            op.residual(rhs, op.block_vector(0.5 * src_u, src_p), None, op.block_vector(old_u), op.block_vector(oldold_u))
            vsrc, vdst = op.initialize_u_vector(vm_u), op.initialize_u_vector()
            op.velocity_vmult(vdst, vsrc)
            assert rel_l2(vdst.numpy(), ref_vel) < TOL, variant
            op.set_kernel_variant(0)                      # generic kernel on the frozen streaming copy
            op.velocity_vmult(vdst, vsrc)
            assert rel_l2(vdst.numpy(), ref_vel) < TOL, variant


@pytest.mark.parametrize("ncell,upper,faces_u", [((9, 8, 5), (1., 1., 1.), range(6)), ((17, 9, 6), (1., 1., 3.), [0, 3, 4]),
                                                  ((5, 4, 9), (1., 2., 1.), range(6)), ((1, 1, 1), (1., 1., 1.), range(6))])
def test_vmult_recomputes_the_state_from_the_nodal_linearisation_point(ncell, upper, faces_u):
    """default since round 5 (kernel variant 1): the Q2/Q1 Newton vmult does not stream (u_lin, grad u_lin) but evaluates
    it from the nodal solution the last residual was computed at (navier_stokes_matrix.cc:778-816: the state IS that
    interpolation); against the oracle's vmult on the oracle's state, after a second residual at another point, with
    the streaming kernel forced (variant 4), and with a state set through the canonical array (no nodal field known: the
    streaming kernel takes over by itself)"""
    case = Case(ncell, k=2, lower=(0., 0., 0.), upper=upper, faces_u=faces_u, steps=3, tau_grad_div=0.1)
    old_u, oldold_u = case.smooth_u(0.05), case.smooth_u(0.0)
    vm_u, vm_p = case.random_u(), case.random_p()
    w, modes = case.weights_modes()
    op = case.engine()
    dst = op.block_vector()
    for scale in (1.0, 0.6):
        src_u, src_p = scale * (case.smooth_u(0.1) + 0.05 * case.random_u()), case.smooth_p(0.1)
        lin_ref = np.zeros(case.n_cells * case.nq * 12)
        orc.ns_residual(case.mesh, case.k, case.prm, src_u, src_p, old_u, oldold_u, con_u=case.con_u, con_p=case.con_p,
                        lin=lin_ref)
        ref_u, ref_p = orc.ns_vmult(case.mesh, 2, case.prm, vm_u, vm_p, case.con_u, case.con_p, lin=lin_ref,
                                    weights=w, modes=modes)
        rhs = op.block_vector()
        op.residual(rhs, op.block_vector(src_u, src_p), None, op.block_vector(old_u), op.block_vector(oldold_u))
        for variant in (1, 4):
            op.set_kernel_variant(variant)
            op.vmult(dst, op.block_vector(vm_u, vm_p))
            gu, gp = dst.numpy()
            assert rel_l2(gu, ref_u) < TOL and rel_l2(gp, ref_p) < TOL, (scale, variant, rel_l2(gu, ref_u), rel_l2(gp, ref_p))
        op.set_kernel_variant(1)
    lin = case.random_lin()
    ref_u, ref_p = orc.ns_vmult(case.mesh, 2, case.prm, vm_u, vm_p, case.con_u, case.con_p, lin=lin, weights=w, modes=modes)
    op.set_linearization(lin)
    op.vmult(dst, op.block_vector(vm_u, vm_p))
    gu, gp = dst.numpy()
    assert rel_l2(gu, ref_u) < TOL and rel_l2(gp, ref_p) < TOL


@pytest.mark.parametrize("ncell,upper,phys,faces_u,lin", [((9, 8, 5), (1., 1., 1.), 0, range(6), 0), ((17, 9, 6), (1., 1., 3.), 0, [0, 3, 4], 0),
                                                           ((5, 4, 9), (1., 2., 1.), 1, range(6), 0), ((1, 1, 1), (1., 1., 1.), 0, range(6), 0),
                                                           ((9, 8, 5), (1., 1., 1.), 0, range(6), 1), ((17, 9, 6), (1., 1., 3.), 0, [0, 3, 4], 1),
                                                           ((9, 8, 5), (1., 1., 1.), 0, range(6), 2), ((17, 9, 6), (1., 1., 3.), 0, [0, 3, 4], 2),
                                                           ((9, 8, 5), (1., 1., 1.), 0, range(6), 3), ((5, 4, 9), (1., 2., 1.), 0, [0, 3, 4], 4)])
def test_two_phase_residual_sweep_kernel(ncell, upper, phys, faces_u, lin):
    """the residual of two-phase flow (variable density / viscosity / damping, navier_stokes_matrix.cc:266-293, 636-642,
    711-713, 831-845) on the Q2/Q1 sweep kernel (template RES with VARCO, round 5): right-hand side with the
    read-modify-write semantics of the reference, the state it leaves (canonical array), the Jacobian on that state in
    the recompute-state mode and -- after a change of kernel variant -- streamed with the coefficient pieces, the frozen
    operator; against the oracle, and the generic kernels on the same inputs.  Round 6: the Picard-type scheme as well (lin = 1:
    its Jacobian recomputes (u_lin, div u_lin) from the nodal field like the Newton one) and, second half, the schemes that
    linearise about the extrapolated old velocity (2 semi-implicit, 3 explicit: no state, 4 projection; rising_bubble_ls_imex /
    _expl of the reference)"""
    case = Case(ncell, k=2, lower=(0., 0., 0.), upper=upper, faces_u=faces_u, physical_type=phys, steps=3,
                tau_grad_div=0.1, density_diff=0.5, linearization=lin)
    src_u, src_p = case.smooth_u(0.1) + 0.05 * case.random_u(), case.smooth_p(0.1)
    old_u, oldold_u = case.smooth_u(0.05), case.smooth_u(0.0)
    rhs0_u, rhs0_p, usr_u, usr_p = case.random_u(), case.random_p(), case.random_u(), case.random_p()
    rho, mu, damp = case.random_coefficients()
    lin_ref = np.zeros(case.n_cells * case.nq * 12)
    ref_u, ref_p = orc.ns_residual(case.mesh, case.k, case.prm, src_u, src_p, old_u, oldold_u, con_u=case.con_u,
                                   con_p=case.con_p, lin=lin_ref, rhs_u=rhs0_u, rhs_p=rhs0_p, user_u=usr_u, user_p=usr_p,
                                   rho=rho, mu=mu, damp=damp)
    vm_u, vm_p = case.random_u(), case.random_p()
    w, modes = case.weights_modes()
    ref_vu, ref_vp = orc.ns_vmult(case.mesh, 2, case.prm, vm_u, vm_p, case.con_u, case.con_p, lin=lin_ref, rho=rho, mu=mu,
                                  damp=damp, weights=w, modes=modes)
    ref_vel = orc.ns_velocity_vmult(case.mesh, 2, case.prm, vm_u, case.con_u, lin=lin_ref, rho=rho, mu=mu, damp=damp)
    for variant in (1, 0):
        op = case.engine()
        op.set_kernel_variant(variant)
        op.set_coefficients(rho, mu, damp)
        rhs = op.block_vector(rhs0_u, rhs0_p)
        op.residual(rhs, op.block_vector(src_u, src_p), op.block_vector(usr_u, usr_p), op.block_vector(old_u),
                    op.block_vector(oldold_u))
        got_u, got_p = rhs.numpy()
        assert rel_l2(got_u, ref_u) < TOL and rel_l2(got_p, ref_p) < TOL, (variant, rel_l2(got_u, ref_u), rel_l2(got_p, ref_p))
        dst = op.block_vector()
        op.vmult(dst, op.block_vector(vm_u, vm_p))
        gu, gp = dst.numpy()
        assert rel_l2(gu, ref_vu) < TOL and rel_l2(gp, ref_vp) < TOL, (variant, rel_l2(gu, ref_vu))
        ncomp = 12 if lin == 0 else 4
        if lin != 3:
            got_lin = op.get_linearization().reshape(-1, 12)
            assert rel_l2(got_lin[:, :ncomp], lin_ref.reshape(-1, 12)[:, :ncomp]) < TOL, variant
        if variant == 1:
            op.set_kernel_variant(4)                      # streamed: the state is re-laid out with the coefficient pieces
            op.vmult(dst, op.block_vector(vm_u, vm_p))
            gu, gp = dst.numpy()
            assert rel_l2(gu, ref_vu) < TOL and rel_l2(gp, ref_vp) < TOL, ("streamed", rel_l2(gu, ref_vu))
            op.set_kernel_variant(1)
            op.residual(rhs, op.block_vector(src_u, src_p), None, op.block_vector(old_u), op.block_vector(oldold_u))
        op.fix_linearization_point()
        op.set_coefficients(*case.random_coefficients())
        op.residual(rhs, op.block_vector(0.5 * src_u, src_p), None, op.block_vector(old_u), op.block_vector(oldold_u))
        vsrc, vdst = op.initialize_u_vector(vm_u), op.initialize_u_vector()
        op.velocity_vmult(vdst, vsrc)
        assert rel_l2(vdst.numpy(), ref_vel) < TOL, variant


@pytest.mark.parametrize("k,ncell,lin,phys,chunk", [(4, (5, 4, 9), 0, 0, 0), (3, (6, 5, 5), 0, 0, 2), (5, (3, 2, 3), 0, 0, 0),
                                                    (4, (9, 3, 2), 1, 0, 4), (3, (4, 4, 3), 1, 0, 0), (4, (4, 4, 4), 0, 1, 0),
                                                    (4, (5, 4, 9), 2, 0, 0), (3, (6, 5, 5), 2, 0, 2), (5, (3, 2, 3), 2, 0, 0),
                                                    (4, (4, 4, 5), 3, 0, 1), (5, (3, 2, 3), 3, 0, 0), (4, (3, 5, 2), 4, 0, 0)])
def test_two_phase_residual_x_marching_kernel(k, ncell, lin, phys, chunk):
    """the residual of two-phase flow (variable density / viscosity / damping, navier_stokes_matrix.cc:266-293, 636-642,
    711-713, 717-732, 827-845) on the Q3..Q5 x-marching kernel (template RES with VARCO, round 6): the coefficients arrive as
    a stream of their own and ride along into the state the residual writes, which IS the streaming state of the
    variable-coefficient Jacobian -- right-hand side with the read-modify-write semantics of the reference, the Jacobian on
    that state, the canonical state, the frozen operator (coefficients and state) while a new residual with new
    coefficients replaces the current one, new coefficients on the old state; Newton and Picard-type, stationary; against
    the oracle, and the generic kernels on the same inputs"""
    case = Case(ncell, k=k, lower=(0., 0., 0.), upper=(1., 1.5, 1.), faces_u=[0, 2, 3, 5], faces_p=[1], linearization=lin,
                physical_type=phys, steps=3, tau_grad_div=0.1, damping=0.1, density_diff=0.5)
    src_u, src_p = case.smooth_u(0.1) + 0.05 * case.random_u(), case.smooth_p(0.1)
    old_u, oldold_u = case.smooth_u(0.05), case.smooth_u(0.0)
    rhs0_u, rhs0_p, usr_u, usr_p = case.random_u(), case.random_p(), case.random_u(), case.random_p()
    rho, mu, damp = case.random_coefficients()
    co2 = case.random_coefficients()
    lin_ref = np.zeros(case.n_cells * case.nq * 12)
    ref_u, ref_p = orc.ns_residual(case.mesh, k, case.prm, src_u, src_p, old_u, oldold_u, con_u=case.con_u,
                                   con_p=case.con_p, lin=lin_ref, rhs_u=rhs0_u, rhs_p=rhs0_p, user_u=usr_u, user_p=usr_p,
                                   rho=rho, mu=mu, damp=damp)
    vm_u, vm_p = case.random_u(), case.random_p()
    w, modes = case.weights_modes()
    ref_vu, ref_vp = orc.ns_vmult(case.mesh, k, case.prm, vm_u, vm_p, case.con_u, case.con_p, lin=lin_ref, rho=rho, mu=mu,
                                  damp=damp, weights=w, modes=modes)
    ref_vel = orc.ns_velocity_vmult(case.mesh, k, case.prm, vm_u, case.con_u, lin=lin_ref, rho=rho, mu=mu, damp=damp)
    ref2_u, ref2_p = orc.ns_vmult(case.mesh, k, case.prm, vm_u, vm_p, case.con_u, case.con_p, lin=lin_ref, rho=co2[0], mu=co2[1],
                                  damp=co2[2], weights=w, modes=modes)
    lin3 = np.zeros_like(lin_ref)
    orc.ns_residual(case.mesh, k, case.prm, 0.5 * src_u, src_p, old_u, oldold_u, con_u=case.con_u, con_p=case.con_p, lin=lin3,
                    rho=co2[0], mu=co2[1], damp=co2[2])
    ref3_u, ref3_p = orc.ns_vmult(case.mesh, k, case.prm, vm_u, vm_p, case.con_u, case.con_p, lin=lin3, rho=co2[0], mu=co2[1],
                                  damp=co2[2], weights=w, modes=modes)
    ncomp = 12 if lin == 0 else 4
    # (second half of round 6: the schemes that linearise about the extrapolated old velocity -- 2 semi-implicit, 3 explicit:
    # no state, 4 projection -- with variable coefficients: templates RES + EXT + VARCO)
    for variant in (1, 0):
        op = case.engine()
        op.set_kernel_variant(variant)
        op.set_x_chunk(chunk)
        op.set_coefficients(rho, mu, damp)
        rhs = op.block_vector(rhs0_u, rhs0_p)
        op.residual(rhs, op.block_vector(src_u, src_p), op.block_vector(usr_u, usr_p), op.block_vector(old_u),
                    op.block_vector(oldold_u))
        got_u, got_p = rhs.numpy()
        assert rel_l2(got_u, ref_u) < TOL and rel_l2(got_p, ref_p) < TOL, (variant, rel_l2(got_u, ref_u), rel_l2(got_p, ref_p))
        dst = op.block_vector()
        op.vmult(dst, op.block_vector(vm_u, vm_p))              # streams what the residual wrote, coefficient pieces included
        gu, gp = dst.numpy()
        assert rel_l2(gu, ref_vu) < TOL and rel_l2(gp, ref_vp) < TOL, (variant, rel_l2(gu, ref_vu))
        if lin != 3:
            got_lin = op.get_linearization().reshape(-1, 12)
            assert rel_l2(got_lin[:, :ncomp], lin_ref.reshape(-1, 12)[:, :ncomp]) < TOL, variant
        op.vmult(dst, op.block_vector(vm_u, vm_p))              # (the streaming copy is still the current one)
        gu, gp = dst.numpy()
        assert rel_l2(gu, ref_vu) < TOL and rel_l2(gp, ref_vp) < TOL, variant
        op.fix_linearization_point()                            # freezes state AND coefficients
        op.set_coefficients(*co2)                               # new coefficients on the old state
        op.vmult(dst, op.block_vector(vm_u, vm_p))
        gu, gp = dst.numpy()
        assert rel_l2(gu, ref2_u) < TOL and rel_l2(gp, ref2_p) < TOL, (variant, rel_l2(gu, ref2_u))
        op.residual(rhs, op.block_vector(0.5 * src_u, src_p), None, op.block_vector(old_u), op.block_vector(oldold_u))
        vsrc, vdst = op.initialize_u_vector(vm_u), op.initialize_u_vector()
        op.velocity_vmult(vdst, vsrc)                           # the frozen operator
        assert rel_l2(vdst.numpy(), ref_vel) < TOL, (variant, rel_l2(vdst.numpy(), ref_vel))
        op.vmult(dst, op.block_vector(vm_u, vm_p))              # the new one
        gu, gp = dst.numpy()
        assert rel_l2(gu, ref3_u) < TOL and rel_l2(gp, ref3_p) < TOL, (variant, rel_l2(gu, ref3_u))
        op.set_kernel_variant(0)                                # generic kernels on the copies the sweep kernel left
        op.velocity_vmult(vdst, vsrc)
        assert rel_l2(vdst.numpy(), ref_vel) < TOL, variant
        op.vmult(dst, op.block_vector(vm_u, vm_p))
        gu, gp = dst.numpy()
        assert rel_l2(gu, ref3_u) < TOL and rel_l2(gp, ref3_p) < TOL, variant


def test_lazy_state_of_the_newton_residual():
    """adaflo_set_q2_lazy_state (round 6): with the recompute-state vmult as consumer the Q2/Q1 Newton residual does not lay
    out the quadrature-point state (NavierStokesMatrix::residual fills linearized_velocities in its cell loop,
    navier_stokes_matrix.cc:778-799: deferred here, not dropped).  Everything a caller can observe is bitwise what the eager
    residual gives: right-hand side, vmult, velocity_vmult, the canonical state (laid out on demand), the streaming kernel
    after a change of variant, the frozen operator; and the device memory the state would take stays free until somebody asks"""
    import torch
    ncell = (24, 24, 24)
    case = Case(ncell, k=2, lower=(0., 0., 0.), upper=(1., 1., 1.), steps=3, tau_grad_div=0.1)
    src_u, src_p = case.smooth_u(0.1) + 0.05 * case.random_u(), case.smooth_p(0.1)
    old_u, oldold_u = case.smooth_u(0.05), case.smooth_u(0.0)
    vm_u, vm_p = case.random_u(), case.random_p()
    state_bytes = case.n_cells * 27 * 12 * 8
    out, taken = {}, {}
    for lazy in (True, False):
        op = case.engine()
        op.set_lazy_state(lazy)
        rhs, dst = op.block_vector(), op.block_vector()
        vsrc, vdst = op.initialize_u_vector(vm_u), op.initialize_u_vector()
        old, oldold = op.block_vector(old_u), op.block_vector(oldold_u)
        sol, vm = op.block_vector(src_u, src_p), op.block_vector(vm_u, vm_p)
        op.residual(rhs, sol, None, old, oldold)
        op.vmult(dst, vm)
        op.velocity_vmult(vdst, vsrc)
        op.synchronize()
        res = [a.copy() for a in rhs.numpy()] + [a.copy() for a in dst.numpy()] + [vdst.numpy().copy()]
        torch.cuda.synchronize()
        free = torch.cuda.mem_get_info()[0]
        res.append(op.get_linearization().copy())               # laid out now: streaming copy (lazy only) + canonical copy
        taken[lazy] = free - torch.cuda.mem_get_info()[0]
        op.vmult(dst, vm)                                       # (recomputed again: the nodal copy is still current)
        res += [a.copy() for a in dst.numpy()]
        op.residual(rhs, sol, None, old, oldold)                # deferred again
        op.set_kernel_variant(4)                                # the streaming kernel needs the laid-out state
        op.vmult(dst, vm)
        res += [a.copy() for a in dst.numpy()]
        op.set_kernel_variant(1)
        op.residual(rhs, sol, None, old, oldold)
        op.fix_linearization_point()                            # frozen copies are made from the laid-out state
        op.residual(rhs, op.block_vector(0.5 * src_u, src_p), None, old, oldold)
        op.velocity_vmult(vdst, vsrc)
        res.append(vdst.numpy().copy())
        out[lazy] = res
        del rhs, dst, vsrc, vdst, old, oldold, sol, vm
        op.clear()
    for a, b in zip(out[True], out[False]):
        assert np.array_equal(a, b)
    # the lazy residual had not allocated the streaming copy: asking for the state takes that much more device memory
    assert taken[True] - taken[False] >= 0.9 * state_bytes, (taken, state_bytes)
    lin_ref = np.zeros(case.n_cells * case.nq * 12)
    orc.ns_residual(case.mesh, 2, case.prm, src_u, src_p, old_u, oldold_u, con_u=case.con_u, con_p=case.con_p, lin=lin_ref)
    assert rel_l2(out[True][5], lin_ref) < TOL
    ref_vel = orc.ns_velocity_vmult(case.mesh, 2, case.prm, vm_u, case.con_u, lin=lin_ref)
    assert rel_l2(out[True][-1], ref_vel) < TOL


@pytest.mark.parametrize("first_coefficients", [False, True])
def test_new_coefficients_leave_the_state_of_the_sweep_residual_alone(first_coefficients):
    """adaflo_ns_set_coefficients after a sweep-kernel residual (what every two-phase time step does): the state exists
    only as the streaming copy without coefficient pieces and is NOT re-laid out; the Jacobian with the new coefficients
    on that state -- recomputed (variant 1), streamed with coefficient pieces (variant 4: converted on demand), generic
    (variant 0) -- and the canonical state against the oracle"""
    case = Case((9, 8, 5), k=2, lower=(0., 0., 0.), upper=(1., 1., 1.), steps=3, tau_grad_div=0.1, density_diff=0.5)
    src_u, src_p = case.smooth_u(0.1) + 0.05 * case.random_u(), case.smooth_p(0.1)
    old_u, oldold_u = case.smooth_u(0.05), case.smooth_u(0.0)
    vm_u, vm_p = case.random_u(), case.random_p()
    w, modes = case.weights_modes()
    co1, co2 = case.random_coefficients(), case.random_coefficients()
    lin_ref = np.zeros(case.n_cells * case.nq * 12)
    kw1 = dict(rho=co1[0], mu=co1[1], damp=co1[2]) if first_coefficients else {}
    orc.ns_residual(case.mesh, case.k, case.prm, src_u, src_p, old_u, oldold_u, con_u=case.con_u, con_p=case.con_p,
                    lin=lin_ref, **kw1)
    ref_u, ref_p = orc.ns_vmult(case.mesh, 2, case.prm, vm_u, vm_p, case.con_u, case.con_p, lin=lin_ref, rho=co2[0],
                                mu=co2[1], damp=co2[2], weights=w, modes=modes)
    op = case.engine()
    if first_coefficients:
        op.set_coefficients(*co1)
    rhs, dst = op.block_vector(), op.block_vector()
    op.residual(rhs, op.block_vector(src_u, src_p), None, op.block_vector(old_u), op.block_vector(oldold_u))
    op.set_coefficients(*co2)
    for variant in (1, 4, 0, 1):
        op.set_kernel_variant(variant)
        op.vmult(dst, op.block_vector(vm_u, vm_p))
        gu, gp = dst.numpy()
        assert rel_l2(gu, ref_u) < TOL and rel_l2(gp, ref_p) < TOL, (variant, rel_l2(gu, ref_u), rel_l2(gp, ref_p))
    assert rel_l2(op.get_linearization().reshape(-1, 12), lin_ref.reshape(-1, 12)) < TOL


@pytest.mark.parametrize("ncell,upper", [((9, 8, 5), (1., 1., 1.)), ((17, 9, 6), (1., 1., 3.)), ((3, 2, 2), (1., 2., 1.))])
def test_two_phase_vmult_recomputes_the_state_from_the_nodal_linearisation_point(ncell, upper):
    """variable density / viscosity / damping (the two-phase Jacobian): the residual runs on the generic kernel, the
    Q2/Q1 Newton vmult and velocity_vmult recompute (u_lin, grad u_lin) from the nodal copy it left and read rho, mu,
    damping from the generic arrays -- no 64-lane streaming copy is built for them; against the oracle, against the
    streaming kernel (variant 4), and the frozen operator after new coefficients and a new residual"""
    case = Case(ncell, k=2, lower=(0., 0., 0.), upper=upper, steps=3, tau_grad_div=0.1, density_diff=0.5)
    old_u, oldold_u = case.smooth_u(0.05), case.smooth_u(0.0)
    vm_u, vm_p = case.random_u(), case.random_p()
    w, modes = case.weights_modes()
    rho, mu, damp = case.random_coefficients()
    src_u, src_p = case.smooth_u(0.1) + 0.05 * case.random_u(), case.smooth_p(0.1)
    lin_ref = np.zeros(case.n_cells * case.nq * 12)
    orc.ns_residual(case.mesh, case.k, case.prm, src_u, src_p, old_u, oldold_u, con_u=case.con_u, con_p=case.con_p,
                    lin=lin_ref, rho=rho, mu=mu, damp=damp)
    ref_u, ref_p = orc.ns_vmult(case.mesh, 2, case.prm, vm_u, vm_p, case.con_u, case.con_p, lin=lin_ref, rho=rho, mu=mu,
                                damp=damp, weights=w, modes=modes)
    ref_vel = orc.ns_velocity_vmult(case.mesh, 2, case.prm, vm_u, case.con_u, lin=lin_ref, rho=rho, mu=mu, damp=damp)
    op = case.engine()
    op.set_coefficients(rho, mu, damp)
    rhs, dst = op.block_vector(), op.block_vector()
    op.residual(rhs, op.block_vector(src_u, src_p), None, op.block_vector(old_u), op.block_vector(oldold_u))
    for variant in (1, 4):
        op.set_kernel_variant(variant)
        op.vmult(dst, op.block_vector(vm_u, vm_p))
        gu, gp = dst.numpy()
        assert rel_l2(gu, ref_u) < TOL and rel_l2(gp, ref_p) < TOL, (variant, rel_l2(gu, ref_u), rel_l2(gp, ref_p))
    op.set_kernel_variant(1)
    op.fix_linearization_point()
    # the operator moves on: other coefficients, another linearisation point -- the frozen one must not follow
    op.set_coefficients(*case.random_coefficients())
    op.residual(rhs, op.block_vector(0.5 * src_u, src_p), None, op.block_vector(old_u), op.block_vector(oldold_u))
    vsrc, vdst = op.initialize_u_vector(vm_u), op.initialize_u_vector()
    for variant in (1, 4):
        op.set_kernel_variant(variant)
        op.velocity_vmult(vdst, vsrc)
        assert rel_l2(vdst.numpy(), ref_vel) < TOL, variant


@pytest.mark.parametrize("k,lin,two_phase", [(2, 0, False), (2, 0, True), (2, 1, False), (2, 2, False), (2, 3, False),
                                              (3, 2, False), (4, 0, False), (4, 2, False), (5, 0, False), (2, 4, False), (4, 4, False),
                                              (3, 0, True), (4, 0, True), (4, 1, True)])
def test_residual_kernels_on_random_meshes_against_the_generic_kernels(k, lin, two_phase):
    """the residual modes of the sweep kernels (most of them built for 512 registers) on a seeded sweep of small meshes --
    cut tiles in every direction, non-cubic cells, one to a few cell layers -- against the generic kernels of the same
    engine (which the other tests tie to the oracle): sums and the stored state; a broad net for allocation-dependent
    faults like the one the extrapolating residual of round 5 had"""
    rng = np.random.default_rng(100 * k + 10 * lin + int(two_phase))
    nmax = {2: 20, 3: 9, 4: 7, 5: 5}[k]
    for trial in range(6):
        ncell = tuple(int(v) for v in rng.integers(1, nmax + 1, 3))
        upper = tuple(float(v) for v in rng.choice([1.0, 1.5, 2.0], 3)) if trial % 2 else (1., 1., 1.)
        case = Case(ncell, k=k, lower=(0., 0., 0.), upper=upper, faces_u=[0, 2, 3, 5], faces_p=[1], linearization=lin,
                    tau_grad_div=0.2, damping=0.1, density=1.2, steps=3, density_diff=0.5 if two_phase else 0.0)
        src_u, src_p = case.smooth_u(0.1) + 0.05 * case.random_u(), case.smooth_p(0.1)
        old_u, oldold_u = case.smooth_u(0.05) + 0.02 * case.random_u(), case.smooth_u(0.0) + 0.02 * case.random_u()
        coefficients = case.random_coefficients() if two_phase else None
        results = []
        for variant in (1, 0):
            op = case.engine()
            op.set_kernel_variant(variant)
            if two_phase:
                op.set_coefficients(*coefficients)
            rhs = op.block_vector()
            op.residual(rhs, op.block_vector(src_u, src_p), None, op.block_vector(old_u), op.block_vector(oldold_u))
            ru, rp = rhs.numpy()
            lin_state = op.get_linearization() if lin != 3 else np.zeros(1)
            results.append((ru, rp, lin_state))
        (au, ap, al), (bu, bp, bl) = results
        assert rel_l2(au, bu) < TOL and rel_l2(ap, bp) < TOL, (ncell, upper, rel_l2(au, bu), rel_l2(ap, bp))
        if lin != 3:
            ncomp = 12 if lin == 0 else 4
            assert rel_l2(al.reshape(-1, 12)[:, :ncomp], bl.reshape(-1, 12)[:, :ncomp]) < TOL, (ncell, upper)


@pytest.mark.parametrize("k,lin,two_phase", [(2, 0, False), (2, 0, True), (2, 1, False), (2, 2, False), (3, 0, False),
                                              (4, 0, False), (4, 0, True), (5, 0, False), (5, 1, False)])
def test_operator_kernels_on_random_meshes_against_the_generic_kernels(k, lin, two_phase):
    """vmult and velocity_vmult of the sweep kernels on the state a residual of the same context left (Q2/Q1 Newton: the
    recompute-state mode, on cubic and non-cubic cells; the streamed modes otherwise), on a seeded sweep of small meshes
    with cut tiles, against the generic kernels"""
    rng = np.random.default_rng(7 + 100 * k + 10 * lin + int(two_phase))
    nmax = {2: 20, 3: 9, 4: 7, 5: 5}[k]
    for trial in range(5):
        ncell = tuple(int(v) for v in rng.integers(1, nmax + 1, 3))
        upper = tuple(float(v) for v in rng.choice([1.0, 1.5, 2.0], 3)) if trial % 2 else (1., 1., 1.)
        case = Case(ncell, k=k, lower=(0., 0., 0.), upper=upper, faces_u=[0, 2, 3, 5], faces_p=[1], linearization=lin,
                    tau_grad_div=0.2, damping=0.1, density=1.2, steps=3, density_diff=0.5 if two_phase else 0.0)
        src_u, src_p = case.smooth_u(0.1) + 0.05 * case.random_u(), case.smooth_p(0.1)
        old_u, oldold_u = case.smooth_u(0.05), case.smooth_u(0.0)
        vm_u, vm_p = case.random_u(), case.random_p()
        coefficients = case.random_coefficients() if two_phase else None
        results = []
        for variant in (1, 0):
            op = case.engine()
            op.set_kernel_variant(variant)
            if two_phase:
                op.set_coefficients(*coefficients)
            rhs, dst = op.block_vector(), op.block_vector()
            op.residual(rhs, op.block_vector(src_u, src_p), None, op.block_vector(old_u), op.block_vector(oldold_u))
            op.vmult(dst, op.block_vector(vm_u, vm_p))
            du, dp = dst.numpy()
            op.fix_linearization_point()
            vsrc, vdst = op.initialize_u_vector(vm_u), op.initialize_u_vector()
            op.velocity_vmult(vdst, vsrc)
            results.append((du, dp, vdst.numpy()))
        (au, ap, av), (bu, bp, bv) = results
        assert rel_l2(au, bu) < TOL and rel_l2(ap, bp) < TOL and rel_l2(av, bv) < TOL, \
            (ncell, upper, rel_l2(au, bu), rel_l2(ap, bp), rel_l2(av, bv))

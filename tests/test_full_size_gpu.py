"""Full-size parity of the BASELINE configurations on one MI355X:
  config 2  128^3 Q2/Q1 Newton vmult against the oracle's OpenMP restatement (oracle/adaflo_oracle_fast.c,
            itself checked against the naive oracle in tests/test_oracle_kats.py) on the same seeded inputs;
  config 5  64^3 Q4/Q3 stationary driven-cavity operator: wave-private sweep kernel against the same OpenMP
            restatement (degree 4), and against the generic per-cell kernel (independent code) + linearity;
  config 3  the whole 256^3 Q2/Q1 problem (422 M DoF) on ONE GPU: sweep kernel against the generic kernel +
            linearity (marked slow);
  128^3     divergence_vmult_add (the Q2 -> Q1 stencil of the block preconditioner) against the oracle's cell loop and
            two more device kernels."""
import numpy as np
import pytest

from common import Case, rel_l2
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-12


def test_config2_128cubed_against_openmp_oracle():
    n = 128
    case = Case((n, n, n), k=2)
    rng = np.random.default_rng(20260515)
    src_u, src_p = rng.uniform(-1, 1, case.n_u), rng.uniform(-1, 1, case.n_p)
    lin = rng.uniform(-1, 1, case.n_cells * 27 * 12)
    w, modes = case.weights_modes()
    orc.fast_set_threads(orc.usable_cores())
    ref_u, ref_p = orc.fast_ns_vmult(case.mesh, 2, case.prm, src_u, src_p, case.con_u, None, lin=lin,
                                     weights=w, modes=modes)
    op = case.engine()
    op.set_linearization(lin)
    del lin
    dst = op.block_vector()
    op.vmult(dst, op.block_vector(src_u, src_p))
    got_u, got_p = dst.numpy()
    assert rel_l2(got_u, ref_u) < TOL and rel_l2(got_p, ref_p) < TOL, (rel_l2(got_u, ref_u), rel_l2(got_p, ref_p))


def _timed_path_against_openmp_oracle(case, k):
    """The path bench.py times, held against the oracle entry by entry: NavierStokesMatrix::residual at the nodal interpolant
    of the Beltrami field (navier_stokes_matrix.cc:266-293; it leaves the linearisation state, :778-799), then vmult on THAT
    state -- kernel variant 1: Q2/Q1 recomputes the state from the nodal linearisation point (ns_q2_kernel, RCP mode), Q3..Q5
    stream what the residual kernel stored (ns_hox_kernel) --, then velocity_vmult on the frozen copy.  The oracle side:
    orc_fast_ns_residual (OpenMP; right-hand side AND state) and orc_fast_ns_vmult on the oracle's own state."""
    orc.fast_set_threads(orc.usable_cores())
    rng = np.random.default_rng(20260515)
    # (the Beltrami interpolant plus 1 % of noise: at the exact solution the residual is a difference of nearly equal terms
    # and its relative error says nothing about the kernel)
    u0, p0 = case.smooth_u(0.0) + 0.01 * rng.uniform(-1, 1, case.n_u), case.smooth_p(0.0) + 0.01 * rng.uniform(-1, 1, case.n_p)
    old_u, oldold_u = case.smooth_u(-0.05), case.smooth_u(-0.1)
    lin = np.zeros(case.n_cells * case.nq * 12)
    ref_ru, ref_rp = orc.fast_ns_residual(case.mesh, k, case.prm, u0, p0, old_u, oldold_u, con_u=case.con_u, lin=lin)
    src_u, src_p = rng.uniform(-1, 1, case.n_u), rng.uniform(-1, 1, case.n_p)
    w, modes = case.weights_modes()
    ref_u, ref_p = orc.fast_ns_vmult(case.mesh, k, case.prm, src_u, src_p, case.con_u, None, lin=lin, weights=w, modes=modes)
    # velocity_vmult (navier_stokes_matrix.cc:337-382) = the velocity block of the same Jacobian: the naive oracle's own
    # entry on small meshes; at full size vmult of (src_u, 0) restricted to the velocity rows (the same block by linearity)
    if case.n_cells <= 32 ** 3:
        ref_v = orc.ns_velocity_vmult(case.mesh, k, case.prm, src_u, case.con_u, lin=lin)
    else:
        ref_v, _ = orc.fast_ns_vmult(case.mesh, k, case.prm, src_u, np.zeros(case.n_p), case.con_u, None, lin=lin)
    del lin
    op = case.engine()
    op.set_kernel_variant(1)
    rhs = op.block_vector()
    op.residual(rhs, op.block_vector(u0, p0), None, op.block_vector(old_u), op.block_vector(oldold_u))
    got_ru, got_rp = rhs.numpy()
    assert rel_l2(got_ru, ref_ru) < TOL and rel_l2(got_rp, ref_rp) < TOL, (rel_l2(got_ru, ref_ru), rel_l2(got_rp, ref_rp))
    src, dst = op.block_vector(src_u, src_p), op.block_vector()
    op.vmult(dst, src)
    got_u, got_p = dst.numpy()
    assert rel_l2(got_u, ref_u) < TOL and rel_l2(got_p, ref_p) < TOL, (rel_l2(got_u, ref_u), rel_l2(got_p, ref_p))
    op.fix_linearization_point()                     # velocity_vmult on the frozen (nodal / streamed) copy
    vdst = op.initialize_u_vector(np.zeros(case.n_u))
    op.velocity_vmult(vdst, op.initialize_u_vector(src_u))
    assert rel_l2(vdst.numpy(), ref_v) < TOL, rel_l2(vdst.numpy(), ref_v)
    return op


def test_timed_path_config2_128cubed_residual_then_recomputed_vmult_against_openmp_oracle():
    """BASELINE configs[1] exactly as bench.py runs it: 128^3 Q2/Q1, Newton, BDF-2; the vmult kernel is the one whose time
    is the headline -- ns_q2_kernel<0,true,true,false,false,false,true,false>, state recomputed from the nodal field"""
    op = _timed_path_against_openmp_oracle(Case((128, 128, 128), k=2, steps=3), 2)
    ksec, kcount = op.get_kernel_statistics()
    assert kcount > 0


def test_timed_path_128cubed_picard_type_state_recomputed_against_openmp_oracle():
    """the same path with the Picard-type linearisation (round 6: its state (u, div u) is recomputed from the nodal field as
    well -- ns_q2_kernel<1, ..., RCP> -- and the residual defers the layout of the state)"""
    _timed_path_against_openmp_oracle(Case((128, 128, 128), k=2, steps=3, linearization=1), 2)


def test_timed_path_config5_q4_cavity_64cubed_residual_then_vmult_against_openmp_oracle():
    """BASELINE configs[4] as `bench.py --config cavity` runs it: the state the x-marching RESIDUAL kernel stored in the
    streaming layout is what the x-marching vmult kernel streams"""
    _timed_path_against_openmp_oracle(Case((64, 64, 64), k=4, lower=(0., 0., 0.), upper=(1., 1., 3.), physical_type=1,
                                           viscosity=0.01), 4)


@pytest.mark.parametrize("k,ncell", [(2, (17, 9, 6)), (3, (5, 4, 3)), (5, (3, 2, 3))])
def test_timed_path_small_meshes_with_the_naive_oracle_velocity_block(k, ncell):
    _timed_path_against_openmp_oracle(Case(ncell, k=k, steps=3, tau_grad_div=0.1, damping=0.2), k)


def _sweep_vs_generic_and_linearity(case, nq):
    op = case.engine()
    rng = np.random.default_rng(11)
    # linearisation point = Beltrami interpolant seen through the residual kernel (as bench.py)
    u0 = case.smooth_u(0.0)
    tmp = op.block_vector()
    op.residual(tmp, op.block_vector(u0, case.smooth_p(0.0)), None, op.block_vector(u0), op.block_vector())
    del tmp
    x_u, x_p = rng.uniform(-1, 1, case.n_u), rng.uniform(-1, 1, case.n_p)
    y_u, y_p = rng.uniform(-1, 1, case.n_u), rng.uniform(-1, 1, case.n_p)
    x = op.block_vector(x_u, x_p)
    dst = op.block_vector()
    res = {}
    for variant in (1, 0):
        op.set_kernel_variant(variant)
        op.vmult(dst, x)
        res[variant] = dst.numpy()
    assert rel_l2(res[1][0], res[0][0]) < TOL and rel_l2(res[1][1], res[0][1]) < TOL
    op.set_kernel_variant(1)
    ax_u, ax_p = res[1]
    x.block(0).set(y_u)
    x.block(1).set(y_p)
    op.vmult(dst, x)
    ay_u, ay_p = dst.numpy()
    a, b = 0.75, -1.25
    x.block(0).set(a * x_u + b * y_u)
    x.block(1).set(a * x_p + b * y_p)
    op.vmult(dst, x)
    z_u, z_p = dst.numpy()
    assert rel_l2(z_u, a * ax_u + b * ay_u) < TOL and rel_l2(z_p, a * ax_p + b * ay_p) < TOL


def test_config5_q4_cavity_64cubed_against_openmp_oracle():
    """BASELINE configs[4]: Q4/Q3, 64^3 cells on [0,1]x[0,1]x[0,3], `incompressible stationary`, mu = 0.01: the
    high-order sweep kernel against the oracle's OpenMP restatement (adaflo_oracle_fast.c takes any degree)
    on the same seeded inputs, entry by entry"""
    case = Case((64, 64, 64), k=4, lower=(0., 0., 0.), upper=(1., 1., 3.), physical_type=1, viscosity=0.01)
    rng = np.random.default_rng(20260515)
    src_u, src_p = rng.uniform(-1, 1, case.n_u), rng.uniform(-1, 1, case.n_p)
    lin = rng.uniform(-1, 1, case.n_cells * 125 * 12)
    w, modes = case.weights_modes()
    orc.fast_set_threads(orc.usable_cores())
    ref_u, ref_p = orc.fast_ns_vmult(case.mesh, 4, case.prm, src_u, src_p, case.con_u, None, lin=lin,
                                     weights=w, modes=modes)
    op = case.engine()
    op.set_linearization(lin)
    del lin
    dst = op.block_vector()
    op.vmult(dst, op.block_vector(src_u, src_p))
    got_u, got_p = dst.numpy()
    assert rel_l2(got_u, ref_u) < TOL and rel_l2(got_p, ref_p) < TOL, (rel_l2(got_u, ref_u), rel_l2(got_p, ref_p))


def test_config5_q4_cavity_64cubed_properties():
    """BASELINE configs[4]: Q4/Q3, 64^3 cells on [0,1]x[0,1]x[0,3], `incompressible stationary`, mu = 0.01"""
    case = Case((64, 64, 64), k=4, lower=(0., 0., 0.), upper=(1., 1., 3.), physical_type=1, viscosity=0.01)
    _sweep_vs_generic_and_linearity(case, 125)


@pytest.mark.slow
def test_config3_256cubed_on_one_gpu_properties():
    """BASELINE configs[2] without the partition: the whole 256^3 Q2/Q1 mesh (422 M DoF, 43.5 GB of
    Newton state) on one MI355X"""
    case = Case((256, 256, 256), k=2)
    _sweep_vs_generic_and_linearity(case, 27)


def test_divergence_128cubed_against_the_oracle_and_two_more_kernels():
    """divergence_vmult_add at the size of BASELINE configs[1]: the register-marching Q2 -> Q1 stencil (variant 1)
    against the oracle's cell loop on the same seeded inputs (about ten seconds of CPU), and against the divergence mode
    of the sweep kernel (variant 2) and the generic per-cell kernel (variant 0) -- independent implementations"""
    case = Case((128, 128, 128), k=2, faces_u=(0, 3, 4), faces_p=(1,), viscosity=0.37)
    rng = np.random.default_rng(7)
    src_u, base = rng.uniform(-1, 1, case.n_u), rng.uniform(-1, 1, case.n_p)
    op = case.engine()
    res = {}
    for variant in (1, 2, 0):
        op.set_kernel_variant(variant)
        dp = op.initialize_p_vector(base)
        op.divergence_vmult_add(dp, op.initialize_u_vector(src_u), True)
        res[variant] = dp.numpy()
    assert rel_l2(res[1], res[2]) < TOL and rel_l2(res[1], res[0]) < TOL, (rel_l2(res[1], res[2]), rel_l2(res[1], res[0]))
    ref = orc.ns_divergence_vmult_add(case.mesh, 2, case.prm, src_u, base, case.con_u, case.con_p, mu=None,
                                      weight_by_viscosity=True)
    assert rel_l2(res[1], ref) < TOL, rel_l2(res[1], ref)

"""GPU parity tests of the level-set operators (HIP through the C ABI vs the CPU oracle)."""
import numpy as np
import pytest

import adaflo_amd
from adaflo_amd import level_set_okz as lso
from common import rel_l2
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-12


class LSCase:
    def __init__(self, ncell, s, k=2, lower=(0., 0., 0.), upper=(1., 1., 2.), faces=(), seed=3):
        self.s, self.k = s, k
        self.mesh = orc.Mesh.make(list(ncell), lower, upper)
        self.bmesh = adaflo_amd.BrickMesh(list(ncell), lower, upper)
        self.rng = np.random.default_rng(seed)
        self.nn = self.mesh.n_nodes(s)
        self.nq = (2 * s) ** 3
        h = [self.mesh.h[d] for d in range(3)]
        self.eps_used = 1.5 * max(h) / s          # two_phase_base.cc:290-291 with epsilon = 1.5
        self.dt, self.weight, self.w_old, self.w_oo = 0.02, 75.0, -100.0, 25.0
        self.epsilon = 1.5
        self.prm = orc.make_ls_params(s, self.eps_used, min(h), self.dt, self.weight, max(h), self.epsilon)
        self.con = orc.boundary_mask(self.mesh, s, 1, faces=list(faces)) if faces else None
        self.ops = lso.LevelSetOperators(self.bmesh, s, velocity_degree=k, constrained_faces=faces)
        self.ops.set_parameters(self.eps_used, self.dt, self.weight, self.w_old, self.w_oo, self.epsilon)
        self.diag = self.rng.uniform(0.5, 2.0, self.nn)
        if faces:
            self.ops.set_diagonal(self.ops.vector(self.diag))

    def rand(self, blocks=1):
        return self.rng.uniform(-1, 1, self.nn * blocks)

    def rand_q(self):
        return self.rng.uniform(-1, 1, self.mesh.n_cells * self.nq * 3)


@pytest.mark.parametrize("variant", [1, 0])
@pytest.mark.parametrize("s,ncell,faces", [(4, (2, 3, 2), ()), (2, (3, 3, 4), (0, 5)), (1, (4, 4, 4), ()),
                                           (3, (2, 2, 2), (2,)), (4, (5, 9, 3), (1, 2, 4)),
                                           (2, (9, 8, 20), (0, 1, 2, 3, 4, 5)), (1, (33, 17, 40), (3,)),
                                           (1, (1, 1, 1), ()), (2, (1, 2, 1), (0, 5)), (4, (1, 1, 2), ()),
                                           # rows longer than / exactly as long as the 62 owned lanes of a wave
                                           (4, (20, 3, 2), (0, 1)), (2, (31, 2, 3), (2, 3)), (1, (61, 4, 13), (4, 5))])
def test_ls_operator_applications(s, ncell, faces, variant):
    """variant 1: structured Q1 sweep kernel (multi-tile, partial tiles, z-chunks in the larger
    cases); variant 0: generic per-cell kernels"""
    c = LSCase(ncell, s, faces=faces)
    c.ops.set_kernel_variant(variant)
    src = c.rand()
    d = c.ops.vector(np.full(c.nn, 9.0))
    # advection
    adv = lso.LevelSetOKZSolverAdvanceConcentration(c.ops)
    uq = c.rand_q()
    adv.evaluated_convection = uq
    adv.advance_concentration_vmult(d, c.ops.vector(src))
    assert rel_l2(d.numpy(), orc.ls_advect_vmult(c.mesh, c.prm, src, uq, con=c.con, diag=c.diag)) < TOL
    assert rel_l2(adv.evaluated_convection, uq) == 0.0
    # reinitialization
    rei = lso.LevelSetOKZSolverReinitialization(c.ops)
    nq = c.rand_q()
    rei.evaluated_normal = nq
    for diffuse_only in (False, True):
        rei.reinitialization_vmult(d, c.ops.vector(src), diffuse_only)
        ref = orc.ls_reinit_vmult(c.mesh, c.prm, src, nq, diffuse_only=diffuse_only, con=c.con, diag=c.diag)
        assert rel_l2(d.numpy(), ref) < TOL
    # normal (3 blocks) and curvature
    src3 = c.rand(3)
    d3 = c.ops.vector(blocks=3)
    lso.LevelSetOKZSolverComputeNormal(c.ops).compute_normal_vmult(d3, c.ops.vector(src3, blocks=3))
    assert rel_l2(d3.numpy(), orc.ls_normal_vmult(c.mesh, c.prm, src3, con=c.con, diag=c.diag)) < TOL
    cur = lso.LevelSetOKZSolverComputeCurvature(c.ops)
    for apply_diffusion in (True, False):
        cur.compute_curvature_vmult(d, c.ops.vector(src), apply_diffusion)
        ref = orc.ls_curvature_vmult(c.mesh, c.prm, src, apply_diffusion=apply_diffusion, con=c.con, diag=c.diag)
        assert rel_l2(d.numpy(), ref) < TOL


@pytest.mark.parametrize("variant", [1, 0])
@pytest.mark.parametrize("s,ncell,faces", [(4, (2, 2, 3), ()), (2, (3, 4, 3), ()), (2, (9, 5, 20), (0, 3, 4, 5)),
                                           (1, (20, 17, 35), (1,)), (1, (1, 1, 1), ()), (2, (1, 2, 1), (2,))])
def test_ls_right_hand_sides(s, ncell, faces, variant):
    """variant 1: normal / curvature right-hand sides as tensor-product stencils (multi-block planes,
    several z-chunks in the larger cases); variant 0: generic per-cell kernels"""
    c = LSCase(ncell, s, faces=faces)
    c.ops.set_kernel_variant(variant)
    phi = c.rand()
    normal = c.rand(3)
    normal[::7] *= 1e-3   # some nearly vanishing normals (thresholds 1e-4 / 1e-2)
    rei = lso.LevelSetOKZSolverReinitialization(c.ops)
    # reinit rhs, first step (writes evaluated_normal), then later step (reads it)
    nq_ref = np.zeros(c.mesh.n_cells * c.nq * 3)
    ref = orc.ls_reinit_rhs(c.mesh, c.prm, phi, normal, nq_ref, diffuse_only=False, first_step=True, con=c.con)
    d = c.ops.vector()
    rei.local_reinitialize_rhs(d, c.ops.vector(phi), c.ops.vector(normal, blocks=3), False, True)
    assert rel_l2(d.numpy(), ref) < TOL
    assert rel_l2(rei.evaluated_normal, nq_ref) < TOL
    phi2 = c.rand()
    ref = orc.ls_reinit_rhs(c.mesh, c.prm, phi2, normal, nq_ref, diffuse_only=False, first_step=False, con=c.con)
    d = c.ops.vector()
    rei.local_reinitialize_rhs(d, c.ops.vector(phi2), None, False, False)
    assert rel_l2(d.numpy(), ref) < TOL
    ref = orc.ls_reinit_rhs(c.mesh, c.prm, phi2, normal, nq_ref, diffuse_only=True, first_step=False, con=c.con)
    d = c.ops.vector()
    rei.local_reinitialize_rhs(d, c.ops.vector(phi2), None, True, False)
    assert rel_l2(d.numpy(), ref) < TOL
    # normal rhs
    d3 = c.ops.vector(blocks=3)
    lso.LevelSetOKZSolverComputeNormal(c.ops).local_compute_normal_rhs(d3, c.ops.vector(phi))
    assert rel_l2(d3.numpy(), orc.ls_normal_rhs(c.mesh, c.prm, phi, con=c.con)) < TOL
    # curvature rhs (with a region of zero normal: early-out cells)
    normal_z = normal.reshape(3, -1).copy()
    normal_z[:, : c.nn // 3] = 0.0
    normal_z = normal_z.reshape(-1)
    d = c.ops.vector()
    lso.LevelSetOKZSolverComputeCurvature(c.ops).local_compute_curvature_rhs(d, c.ops.vector(normal_z, blocks=3))
    assert rel_l2(d.numpy(), orc.ls_curvature_rhs(c.mesh, c.prm, normal_z, con=c.con)) < TOL
    # advection rhs
    adv = lso.LevelSetOKZSolverAdvanceConcentration(c.ops)
    vel = c.rng.uniform(-1, 1, c.mesh.n_nodes(c.k) * 3)
    old, oldold = c.rand(), c.rand()
    for use_oo in (True, False):
        uq_ref = np.zeros(c.mesh.n_cells * c.nq * 3)
        ref = orc.ls_advect_rhs(c.mesh, c.prm, c.k, phi, old, oldold, vel, uq_ref, c.w_old, c.w_oo, use_oo, con=c.con)
        d = c.ops.vector()
        adv.local_advance_concentration_rhs(d, c.ops.vector(phi), c.ops.vector(old), c.ops.vector(oldold),
                                            c.ops.velocity_vector(vel), use_oo)
        assert rel_l2(d.numpy(), ref) < TOL
        assert rel_l2(adv.evaluated_convection, uq_ref) < TOL


@pytest.mark.parametrize("s,ncell,faces", [(4, (5, 4, 3), ()), (2, (9, 17, 6), (0, 3)), (1, (20, 18, 35), (4, 5)), (3, (1, 1, 1), ())])
def test_reinitialization_vmult_recomputes_the_normal_from_the_nodal_field(s, ncell, faces):
    """after a first-step reinitialisation rhs the engine keeps the nodal normal field and the reinitialisation vmult
    recomputes the unit normal at the Gauss points (Q1_REINIT_NODAL, csrc/q1_sweep.hip) instead of streaming
    evaluated_normal: same result as the oracle fed with the evaluated_normal the rhs produced, and as the streaming
    kernel once evaluated_normal is set explicitly"""
    c = LSCase(ncell, s, faces=faces)
    c.ops.set_kernel_variant(1)
    rei = lso.LevelSetOKZSolverReinitialization(c.ops)
    phi, nrm = c.ops.vector(c.rand()), c.ops.vector(c.rand(3), blocks=3)
    rhs = c.ops.vector()
    rei.local_reinitialize_rhs(rhs, phi, nrm, False, True)
    src = c.rand()
    d = c.ops.vector(np.full(c.nn, 9.0))
    rei.reinitialization_vmult(d, c.ops.vector(src), False)          # nodal path
    got_nodal = d.numpy().copy()
    nq = np.array(rei.evaluated_normal)                              # what the rhs kernel stored per Gauss point
    ref = orc.ls_reinit_vmult(c.mesh, c.prm, src, nq, diffuse_only=False, con=c.con, diag=c.diag)
    assert rel_l2(got_nodal, ref) < TOL
    rei.evaluated_normal = nq                                        # explicit state: streaming path
    rei.reinitialization_vmult(d, c.ops.vector(src), False)
    assert rel_l2(d.numpy(), got_nodal) < 1e-13


@pytest.mark.parametrize("s,k,ncell,faces", [(4, 2, (5, 4, 3), ()), (3, 2, (7, 6, 5), (0, 3)), (2, 3, (9, 5, 6), ()), (4, 4, (5, 3, 2), (4, 5)),
                                             (1, 2, (20, 18, 9), (1,)), (2, 2, (1, 1, 1), ())])
def test_advection_vmult_evaluates_the_velocity_from_the_nodal_field(s, k, ncell, faces, monkeypatch):
    """after an advection rhs on the sweep structure the engine keeps the nodal velocity instead of writing
    evaluated_convection (192 B per sub-cell) and the advection vmult evaluates the FE_Q(k) velocity at the Gauss points
    itself (Q1_ADVECT_NODAL, csrc/q1_sweep.hip): same result as the oracle fed with the evaluated_convection of ITS rhs,
    as the streaming kernel (ADAFLO_LS_STREAM_CONVECTION: the rhs writes the array, the operator streams it), and as the
    streaming kernel on the array the engine materialises on demand"""
    c = LSCase(ncell, s, k=k, faces=faces)
    c.ops.set_kernel_variant(1)
    adv = lso.LevelSetOKZSolverAdvanceConcentration(c.ops)
    phi, old, oldold = c.rand(), c.rand(), c.rand()
    vel = c.rng.uniform(-1, 1, c.mesh.n_nodes(k) * 3)
    uq_ref = np.zeros(c.mesh.n_cells * c.nq * 3)
    ref_rhs = orc.ls_advect_rhs(c.mesh, c.prm, k, phi, old, oldold, vel, uq_ref, c.w_old, c.w_oo, True, con=c.con)
    src = c.rand()
    ref = orc.ls_advect_vmult(c.mesh, c.prm, src, uq_ref, con=c.con, diag=c.diag if faces else None)

    def rhs_then_vmult():
        d = c.ops.vector()
        vv = c.ops.velocity_vector(vel)
        adv.local_advance_concentration_rhs(d, c.ops.vector(phi), c.ops.vector(old), c.ops.vector(oldold), vv, True)
        assert rel_l2(d.numpy(), ref_rhs) < TOL
        del vv                                                # (the engine keeps its own copy of the velocity)
        junk = c.ops.velocity_vector(np.full_like(vel, 1e30))  # (likely the same device memory)
        out = c.ops.vector(np.full(c.nn, 7.0))
        adv.advance_concentration_vmult(out, c.ops.vector(src))
        del junk
        return out.numpy().copy()
    got_nodal = rhs_then_vmult()
    assert rel_l2(got_nodal, ref) < TOL
    monkeypatch.setenv("ADAFLO_LS_STREAM_CONVECTION", "1")
    got_stream = rhs_then_vmult()
    monkeypatch.delenv("ADAFLO_LS_STREAM_CONVECTION")
    assert rel_l2(got_stream, ref) < TOL and rel_l2(got_nodal, got_stream) < 1e-13
    # the quadrature-point array on demand (written from the kept velocity), then set explicitly: streaming kernel
    rhs_then_vmult()
    uq = np.array(adv.evaluated_convection)
    assert rel_l2(uq, uq_ref) < TOL
    adv.evaluated_convection = uq
    out = c.ops.vector()
    adv.advance_concentration_vmult(out, c.ops.vector(src))
    assert rel_l2(out.numpy(), got_nodal) < 1e-13


@pytest.mark.parametrize("s,k,ncell,faces", [(3, 2, (7, 6, 5), ()), (3, 3, (6, 7, 3), (1, 2)), (4, 4, (5, 3, 2), ()),
                                             (1, 3, (18, 17, 6), (0,)), (2, 4, (9, 2, 3), (4, 5)),
                                             # velocity degree 5 (level_set_okz_template_instantations.h: 2 .. 5): generic kernels
                                             (2, 5, (3, 2, 3), (1,)), (4, 5, (2, 2, 2), ())])
def test_sweep_right_hand_sides_unaligned_tiles_and_velocity_degrees(s, k, ncell, faces):
    """advection / reinitialisation right-hand sides on the sweep structure (csrc/q1_sweep.hip) where the 16 x 16
    sub-cell tiles are not aligned with the cells (s = 3) and for velocity degrees 3 and 4: the velocity patch of
    a tile starts inside a cell, several tiles and z-chunks; evaluated_* read back through the generic layout"""
    c = LSCase(ncell, s, k=k, faces=faces)
    phi, old, oldold, normal = c.rand(), c.rand(), c.rand(), c.rand(3)
    vel = c.rng.uniform(-1, 1, c.mesh.n_nodes(k) * 3)
    adv, rei = lso.LevelSetOKZSolverAdvanceConcentration(c.ops), lso.LevelSetOKZSolverReinitialization(c.ops)
    for use_oo in (True, False):
        uq_ref = np.zeros(c.mesh.n_cells * c.nq * 3)
        base = c.rand()                                     # the kernels ADD into dst
        ref = base + orc.ls_advect_rhs(c.mesh, c.prm, k, phi, old, oldold, vel, uq_ref, c.w_old, c.w_oo, use_oo, con=c.con)
        d = c.ops.vector(base)
        adv.local_advance_concentration_rhs(d, c.ops.vector(phi), c.ops.vector(old), c.ops.vector(oldold),
                                            c.ops.velocity_vector(vel), use_oo)
        assert rel_l2(d.numpy(), ref) < TOL
        assert rel_l2(adv.evaluated_convection, uq_ref) < TOL
    vmax = adv.get_maximal_velocity(c.ops.velocity_vector(vel))
    assert abs(vmax - orc.ls_max_velocity(c.mesh, k, vel)) < 1e-13 * vmax
    # the operator reads the state the right-hand side left in sweep layout
    src, dst = c.rand(), c.ops.vector()
    adv.advance_concentration_vmult(dst, c.ops.vector(src))
    ref = orc.ls_advect_vmult(c.mesh, c.prm, src, uq_ref, con=c.con, diag=c.diag if faces else None)
    assert rel_l2(dst.numpy(), ref) < TOL
    nq_ref = np.zeros(c.mesh.n_cells * c.nq * 3)
    for first in (True, False):
        base = c.rand()
        ref = base + orc.ls_reinit_rhs(c.mesh, c.prm, phi, normal, nq_ref, diffuse_only=False, first_step=first, con=c.con)
        d = c.ops.vector(base)
        rei.local_reinitialize_rhs(d, c.ops.vector(phi), c.ops.vector(normal, blocks=3) if first else None, False, first)
        assert rel_l2(d.numpy(), ref) < TOL
    assert rel_l2(rei.evaluated_normal, nq_ref) < TOL
    base = c.rand()
    ref = base + orc.ls_reinit_rhs(c.mesh, c.prm, phi, normal, nq_ref, diffuse_only=True, first_step=False, con=c.con)
    d = c.ops.vector(base)
    rei.local_reinitialize_rhs(d, c.ops.vector(phi), None, True, False)
    assert rel_l2(d.numpy(), ref) < TOL


def test_config4_full_size_against_oracle():
    """BASELINE configs[3] at full size (40 x 40 x 80 cells, s = 4: 8.3 M level-set DoF, 65.5 M quadrature points):
    the sweep kernels of advance_concentration_vmult, reinitialization_vmult and of the two right-hand sides, and
    the 27-point stencil kernel of the normal / curvature operators,
    against the CPU oracle's cell loops on identical inputs, entry by entry.  The six oracle evaluations
    (naive restatement, tens of seconds each) run concurrently in threads."""
    from concurrent.futures import ThreadPoolExecutor
    c = LSCase((40, 40, 80), 4)
    src, phi, old, oldold, normal = c.rand(), c.rand(), c.rand(), c.rand(), c.rand(3)
    uq, nq = c.rand_q(), c.rand_q()
    vel = c.rng.uniform(-1, 1, c.mesh.n_nodes(c.k) * 3)
    uq_ref, nq_ref = np.zeros_like(uq), np.zeros_like(nq)
    src3 = c.rand(3)
    with ThreadPoolExecutor(6) as pool:          # (ctypes calls release the interpreter lock)
        f_nor = pool.submit(orc.ls_normal_vmult, c.mesh, c.prm, src3)
        f_cur = pool.submit(orc.ls_curvature_vmult, c.mesh, c.prm, src, apply_diffusion=True)
        f_adv = pool.submit(orc.ls_advect_vmult, c.mesh, c.prm, src, uq)
        f_rei = pool.submit(orc.ls_reinit_vmult, c.mesh, c.prm, src, nq, diffuse_only=False)
        f_arhs = pool.submit(orc.ls_advect_rhs, c.mesh, c.prm, c.k, phi, old, oldold, vel, uq_ref, c.w_old, c.w_oo, True)
        f_rrhs = pool.submit(orc.ls_reinit_rhs, c.mesh, c.prm, phi, normal, nq_ref, diffuse_only=False, first_step=True)
        # the device side meanwhile
        adv, rei = lso.LevelSetOKZSolverAdvanceConcentration(c.ops), lso.LevelSetOKZSolverReinitialization(c.ops)
        d = c.ops.vector()
        adv.evaluated_convection = uq
        adv.advance_concentration_vmult(d, c.ops.vector(src))
        got_adv = d.numpy()
        rei.evaluated_normal = nq
        rei.reinitialization_vmult(d, c.ops.vector(src), False)
        got_rei = d.numpy()
        d = c.ops.vector()
        adv.local_advance_concentration_rhs(d, c.ops.vector(phi), c.ops.vector(old), c.ops.vector(oldold),
                                            c.ops.velocity_vector(vel), True)
        got_arhs, got_uq = d.numpy(), adv.evaluated_convection
        d = c.ops.vector()
        rei.local_reinitialize_rhs(d, c.ops.vector(phi), c.ops.vector(normal, blocks=3), False, True)
        got_rrhs, got_nq = d.numpy(), rei.evaluated_normal
        # after the first-step rhs the operator recomputes the normal from the nodal field (Q1_REINIT_NODAL); with the
        # state set explicitly it streams it: both at full size
        d = c.ops.vector()
        rei.reinitialization_vmult(d, c.ops.vector(src), False)
        got_nodal = d.numpy()
        rei.evaluated_normal = np.array(got_nq)
        rei.reinitialization_vmult(d, c.ops.vector(src), False)
        assert rel_l2(got_nodal, d.numpy()) < 1e-13, "reinitialization_vmult: nodal normal vs streamed state"
        # the 27-point stencil kernel: normal (three blocks) and curvature operators
        d3 = c.ops.vector(blocks=3)
        lso.LevelSetOKZSolverComputeNormal(c.ops).compute_normal_vmult(d3, c.ops.vector(src3, blocks=3))
        got_nor = d3.numpy()
        lso.LevelSetOKZSolverComputeCurvature(c.ops).compute_curvature_vmult(d, c.ops.vector(src), True)
        got_cur = d.numpy()
        assert rel_l2(got_nor, f_nor.result()) < TOL, "compute_normal_vmult vs oracle"
        assert rel_l2(got_cur, f_cur.result()) < TOL, "compute_curvature_vmult vs oracle"
        assert rel_l2(got_adv, f_adv.result()) < TOL, "advance_concentration_vmult vs oracle"
        assert rel_l2(got_rei, f_rei.result()) < TOL, "reinitialization_vmult vs oracle"
        assert rel_l2(got_arhs, f_arhs.result()) < TOL and rel_l2(got_uq, uq_ref) < TOL, "advection rhs vs oracle"
        assert rel_l2(got_rrhs, f_rrhs.result()) < TOL and rel_l2(got_nq, nq_ref) < TOL, "reinitialisation rhs vs oracle"


def test_full_size_properties_config4():
    """Config 4 (40x40x80 cells, s = 4, 8.3 M level-set DoF): the structured Q1 sweep kernel and
    the generic per-cell kernels (independent code) agree for every operator application, and
    the curvature operator (mass + damping Laplacian, no constraints) is symmetric."""
    c = LSCase((40, 40, 80), 4)
    x, y = c.rand(), c.rand()
    uq, nq = c.rand_q(), c.rand_q()
    adv = lso.LevelSetOKZSolverAdvanceConcentration(c.ops)
    rei = lso.LevelSetOKZSolverReinitialization(c.ops)
    nor = lso.LevelSetOKZSolverComputeNormal(c.ops)
    cur = lso.LevelSetOKZSolverComputeCurvature(c.ops)
    adv.evaluated_convection = uq
    rei.evaluated_normal = nq
    xv, x3 = c.ops.vector(x), c.ops.vector(np.concatenate([x, y, x - y]), blocks=3)
    out = {}
    for variant in (1, 0):
        c.ops.set_kernel_variant(variant)
        d, d3 = c.ops.vector(), c.ops.vector(blocks=3)
        adv.advance_concentration_vmult(d, xv)
        r = [d.numpy()]
        rei.reinitialization_vmult(d, xv, False)
        r.append(d.numpy())
        rei.reinitialization_vmult(d, xv, True)
        r.append(d.numpy())
        nor.compute_normal_vmult(d3, x3)
        r.append(d3.numpy())
        cur.compute_curvature_vmult(d, xv, True)
        r.append(d.numpy())
        out[variant] = r
    for a, b in zip(out[1], out[0]):
        assert rel_l2(a, b) < TOL
    c.ops.set_kernel_variant(1)
    d = c.ops.vector()
    cur.compute_curvature_vmult(d, c.ops.vector(y), True)
    ay = d.numpy()
    ax = out[1][4]
    assert abs(y @ ax - x @ ay) < 1e-12 * np.linalg.norm(ax) * np.linalg.norm(y)


@pytest.mark.parametrize("s,ncell,faces", [(4, (3, 2, 3), ()), (2, (5, 4, 9), (0, 3, 5))])
def test_projection_matrix_and_curvature_correction(s, ncell, faces):
    """adaflo_ls_projection_vmult = one scalar block of the normal operator (the assembled projection
    matrix of level_set_okz.cc:262-312); adaflo_ls_curvature_correction = compute_curvature.cc:360-376"""
    from adaflo_amd import _lib
    c = LSCase(ncell, s, faces=faces)
    lib, ctx = _lib.load(), c.ops._ctx
    src = c.rand()
    d = c.ops.vector(np.full(c.nn, 7.0))
    _lib.check(ctx, lib.adaflo_ls_projection_vmult(ctx, d.ptr, c.ops.vector(src).ptr))
    blocks = np.concatenate([src, np.zeros(2 * c.nn)])
    ref = orc.ls_normal_vmult(c.mesh, c.prm, blocks, con=c.con, diag=c.diag)[:c.nn]
    assert rel_l2(d.numpy(), ref) < TOL
    # curvature correction: kappa > 1e-4 -> 1 / (1 / kappa + distance / 2), distance from the level set
    kappa = c.rng.uniform(-3.0, 8.0, c.nn)
    kappa[::5] = 5e-5
    phi = np.tanh(c.rng.uniform(-4.0, 4.0, c.nn))
    kv = c.ops.vector(kappa)
    _lib.check(ctx, lib.adaflo_ls_curvature_correction(ctx, kv.ptr, c.ops.vector(phi).ptr))
    with np.errstate(divide="ignore", invalid="ignore"):
        dist = np.where(1 - phi * phi > 1e-2, c.eps_used * np.log((1 + phi) / (1 - phi)), 0.0)
    expect = kappa.copy()
    sel = kappa > 1e-4
    expect[sel] = 1.0 / (1.0 / kappa[sel] + dist[sel] / 2.0)
    assert np.abs(kv.numpy() - expect).max() < 1e-13 * max(1.0, np.abs(expect).max())


@pytest.mark.parametrize("s,k,ncell,faces,symmetry", [(2, 2, (3, 3, 4), (), ()), (4, 2, (2, 3, 2), (0, 5), (2,)),
                                                      (1, 3, (4, 3, 5), (), (0, 1)), (3, 2, (2, 2, 3), (3,), ())])
def test_advection_with_convection_stabilization(s, k, ncell, faces, symmetry):
    """parameters.convection_stabilization (level_set_okz_advance_concentration.cc): maximal velocity (:39-68),
    artificial viscosity per cell (:344-369), cell terms of the right-hand side (:387-388) and of the operator
    (:248-249), boundary terms (:569-617, :419-472) with symmetry faces left out -- against the oracle"""
    c = LSCase(ncell, s, k=k, faces=faces)
    adv = lso.LevelSetOKZSolverAdvanceConcentration(c.ops)
    adv.set_convection_stabilization(True, symmetry_faces=symmetry)
    sym = sum(1 << f for f in symmetry)
    nv = c.mesh.n_nodes(k) * 3
    vel, vel_o, vel_oo = (c.rng.uniform(-1, 1, nv) for _ in range(3))
    sol, old, oo = c.rand(), c.rand(), c.rand()
    vmax = adv.get_maximal_velocity(c.ops.velocity_vector(vel))
    assert abs(vmax - orc.ls_max_velocity(c.mesh, k, vel)) < 1e-13 * vmax
    old_step = 0.017
    scaling = vmax * 2.0 * adv.global_omega_diameter
    assert abs(adv.global_omega_diameter - np.sqrt(1 + 1 + 4)) < 1e-14
    for use_oo in (True, False):
        uq, nu_ref = np.zeros(c.mesh.n_cells * c.nq * 3), np.zeros(c.mesh.n_cells)
        ref = orc.ls_advect_rhs(c.mesh, c.prm, k, sol, old, oo, vel, uq, c.w_old, c.w_oo, use_oo, con=c.con,
                                vel_old=vel_o, vel_old_old=vel_oo, old_step_size=old_step, global_scaling=scaling,
                                art_visc=nu_ref)
        ref = orc.ls_advect_boundary_term(c.mesh, c.prm, sol, nu_ref, 1.0, ref, con=c.con, symmetry=sym)
        d = c.ops.vector()
        adv.local_advance_concentration_rhs_stabilized(d, c.ops.vector(sol), c.ops.vector(old), c.ops.vector(oo),
                                                       c.ops.velocity_vector(vel), c.ops.velocity_vector(vel_o),
                                                       c.ops.velocity_vector(vel_oo), use_oo, old_step, vmax)
        assert rel_l2(adv.artificial_viscosities, nu_ref) < TOL and nu_ref.min() > 0
        assert rel_l2(d.numpy(), ref) < TOL
        assert rel_l2(adv.evaluated_convection, uq) < 1e-14
    # operator with these viscosities, on both kernel variants (the stabilised operator is a generic kernel)
    src = c.rand()
    ref = orc.ls_advect_vmult(c.mesh, c.prm, src, uq, con=c.con, diag=c.diag, art_visc=nu_ref, symmetry=sym)
    for variant in (1, 0):
        c.ops.set_kernel_variant(variant)
        d = c.ops.vector(np.full(c.nn, 7.0))
        adv.advance_concentration_vmult(d, c.ops.vector(src))
        assert rel_l2(d.numpy(), ref) < TOL, variant
    # a viscosity array set from outside, stabilisation switched off again
    nu2 = c.rng.uniform(0.1, 1.0, c.mesh.n_cells)
    adv.artificial_viscosities = nu2
    adv.advance_concentration_vmult(d, c.ops.vector(src))
    assert rel_l2(d.numpy(), orc.ls_advect_vmult(c.mesh, c.prm, src, uq, con=c.con, diag=c.diag, art_visc=nu2,
                                                 symmetry=sym)) < TOL
    adv.set_convection_stabilization(False)
    adv.advance_concentration_vmult(d, c.ops.vector(src))
    assert rel_l2(d.numpy(), orc.ls_advect_vmult(c.mesh, c.prm, src, uq, con=c.con, diag=c.diag)) < TOL

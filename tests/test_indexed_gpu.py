"""Indexed context (adaflo_ctx_create_indexed; SURVEY 8(b).1, first alternative): the Navier-Stokes block on a Cartesian mesh
that is NOT one brick -- an L-shaped union of bricks with different cell sizes, described by per-cell node tables,
constraint flags, per-cell extents and a colouring -- against the oracle.  The oracle knows bricks only; the operator of
the union is the sum of the cell loops of its bricks (local_operation is additive over cells,
source/navier_stokes_matrix.cc:232-245), so the reference is assembled from one oracle call per brick in the union's
numbering, with the union's constraint flags handed to every call."""
import numpy as np
import pytest

import adaflo_amd
from common import Case, rel_l2
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-12

# lattice 4 x 3 x 2 cells minus the block x >= 2, y >= 1: brick A = [0,4) x [0,1) x [0,2), brick B = [0,2) x [1,3) x [0,2)
HX, HY, HZ = [0.25] * 4, [0.3, 0.4, 0.4], [0.5, 0.5]
BRICKS = [((0, 0, 0), (4, 1, 2)), ((0, 1, 0), (2, 2, 2))]          # (first cell, cells per direction)


class Union:
    def __init__(self, k, **kw):
        self.k = k
        cells = [(a[0] + i, a[1] + j, a[2] + l) for a, n in BRICKS for l in range(n[2]) for j in range(n[1]) for i in range(n[0])]
        self.mesh = adaflo_amd.IndexedMesh(cells, HX, HY, HZ, k)
        self.case = Case((1, 1, 1), k=k, **kw)                          # parameters, time stepping (mesh unused)
        self.table_pos = {tuple(c): i for i, c in enumerate(self.mesh.cells.tolist())}
        self.nq = (k + 1) ** 3
        self.parts = []
        edges = [np.concatenate([[0.0], np.cumsum(h)]) for h in (HX, HY, HZ)]
        for first, n in BRICKS:
            lower = [edges[d][first[d]] for d in range(3)]
            upper = [edges[d][first[d] + n[d]] for d in range(3)]
            omesh = orc.Mesh.make(list(n), lower, upper)
            maps = {}
            for degree in (k, k - 1):
                lat = self.mesh.node_lattice[degree]
                ident = {tuple(p): i for i, p in enumerate(lat.tolist())}
                nn = [degree * n[d] + 1 for d in range(3)]
                maps[degree] = np.array([ident[(degree * first[0] + i, degree * first[1] + j, degree * first[2] + l)]
                                         for l in range(nn[2]) for j in range(nn[1]) for i in range(nn[0])])
            cellpos = np.array([self.table_pos[(first[0] + i, first[1] + j, first[2] + l)]
                                for l in range(n[2]) for j in range(n[1]) for i in range(n[0])])
            self.parts.append((omesh, maps, cellpos))
        self.n_u, self.n_p = 3 * self.mesh.n_nodes(k), self.mesh.n_nodes(k - 1)
        self.flag_u, self.flag_p = self.mesh.constrained_u.astype(bool), self.mesh.constrained_p.astype(bool)

    def engine(self, pressure_average_fix=True):
        op = adaflo_amd.NavierStokesMatrix(self.case.fp, self.mesh)
        op.initialize(self.case.ts, pressure_average_fix)
        return op

    # -- the union's vectors seen by one brick ----------------------------------------------------------------------------------
    def local(self, part, vec_u=None, vec_p=None):
        _, maps, _ = part
        out = []
        if vec_u is not None:
            out.append(vec_u.reshape(-1, 3)[maps[self.k]].reshape(-1).copy())
        if vec_p is not None:
            out.append(vec_p[maps[self.k - 1]].copy())
        return out

    def local_q(self, part, arr, width):
        return None if arr is None else arr.reshape(self.mesh.n_cells, -1)[part[2]].reshape(-1).copy()

    def weights(self):
        w = np.zeros(self.n_p)
        for part in self.parts:
            omesh, maps, _ = part
            np.add.at(w, maps[self.k - 1], orc.ns_pressure_mass_weight(omesh, self.k, self.flag_p[maps[self.k - 1]].astype(np.uint8)))
        return w

    def vmult(self, src_u, src_p, lin, coef=(None, None, None), project=True):
        ref_u, ref_p = np.zeros(self.n_u), np.zeros(self.n_p)
        for part in self.parts:
            omesh, maps, _ = part
            lu, lp = self.local(part, src_u, src_p)
            cu = np.repeat(maps[self.k], 3) * 3 + np.tile(np.arange(3), len(maps[self.k]))
            fu, fp = self.flag_u[cu].astype(np.uint8), self.flag_p[maps[self.k - 1]].astype(np.uint8)
            du, dp = orc.ns_vmult(omesh, self.k, self.case.prm, lu, lp, fu, fp, lin=self.local_q(part, lin, 12 * self.nq),
                                  rho=self.local_q(part, coef[0], self.nq), mu=self.local_q(part, coef[1], self.nq),
                                  damp=self.local_q(part, coef[2], self.nq))
            du[fu.astype(bool)], dp[fp.astype(bool)] = 0., 0.            # (every brick returns +-src there: set once below)
            np.add.at(ref_u, cu, du)
            np.add.at(ref_p, maps[self.k - 1], dp)
        ref_u[self.flag_u], ref_p[self.flag_p] = src_u[self.flag_u], -src_p[self.flag_p]
        if project:
            modes = np.where(self.flag_p, 0., 1.)
            ref_p = orc.ns_pressure_projection(ref_p, self.weights(), modes)
        return ref_u, ref_p


@pytest.mark.parametrize("k", [2, 3])
@pytest.mark.parametrize("lin_scheme,two_phase", [(0, False), (1, False), (0, True)])
def test_vmult_on_an_l_shaped_union_of_bricks(k, lin_scheme, two_phase):
    u = Union(k, linearization=lin_scheme, tau_grad_div=0.1, damping=0.2, density_diff=0.5 if two_phase else 0.0, steps=3)
    rng = np.random.default_rng(7 + k)
    src_u, src_p = rng.uniform(-1, 1, u.n_u), rng.uniform(-1, 1, u.n_p)
    lin = rng.uniform(-1, 1, u.mesh.n_cells * u.nq * 12)
    coef = tuple(rng.uniform(lo, hi, u.mesh.n_cells * u.nq) for lo, hi in ((.5, 2.), (.5, 2.), (-.5, .5))) if two_phase else (None,) * 3
    ref_u, ref_p = u.vmult(src_u, src_p, lin, coef)
    op = u.engine()
    assert op.n_dofs_u() == u.n_u and op.n_dofs_p() == u.n_p and op.n_cells() == u.mesh.n_cells
    op.set_linearization(lin)
    if two_phase:
        op.set_coefficients(*coef)
    dst = op.block_vector(np.full(u.n_u, 7.0), np.full(u.n_p, 7.0))
    op.vmult(dst, op.block_vector(src_u, src_p))
    got_u, got_p = dst.numpy()
    assert rel_l2(got_u, ref_u) < TOL and rel_l2(got_p, ref_p) < TOL, (rel_l2(got_u, ref_u), rel_l2(got_p, ref_p))
    assert np.array_equal(got_u[u.flag_u], src_u[u.flag_u])              # constrained rows: the identity


@pytest.mark.parametrize("k", [2, 4])
def test_residual_state_and_velocity_block_on_the_union(k):
    """NavierStokesMatrix::residual (right-hand side with a user vector, the state it stores in the order of the cell table),
    vmult on that state, velocity_vmult on the frozen copy"""
    u = Union(k, tau_grad_div=0.1, steps=3)
    rng = np.random.default_rng(3)
    X = u.mesh.node_coordinates(k)
    sol_u = (0.3 * np.stack([np.sin(2 * X[:, 0] + X[:, 1]), np.cos(X[:, 1] - X[:, 2]), np.sin(X[:, 2] + 3 * X[:, 0])], axis=1)
             + 0.05 * rng.uniform(-1, 1, (len(X), 3))).reshape(-1)
    sol_p, old_u, oldold_u = rng.uniform(-1, 1, u.n_p), rng.uniform(-1, 1, u.n_u), rng.uniform(-1, 1, u.n_u)
    usr_u, usr_p = rng.uniform(-1, 1, u.n_u), rng.uniform(-1, 1, u.n_p)
    ref_ru, ref_rp, lin = np.zeros(u.n_u), np.zeros(u.n_p), np.zeros((u.mesh.n_cells, u.nq * 12))
    for part in u.parts:
        omesh, maps, cellpos = part
        lu, lp = u.local(part, sol_u, sol_p)
        lo, = u.local(part, old_u)
        loo, = u.local(part, oldold_u)
        cu = np.repeat(maps[k], 3) * 3 + np.tile(np.arange(3), len(maps[k]))
        fu, fp = u.flag_u[cu].astype(np.uint8), u.flag_p[maps[k - 1]].astype(np.uint8)
        l_b = np.zeros(len(cellpos) * u.nq * 12)
        ru, rp = orc.ns_residual(omesh, k, u.case.prm, lu, lp, lo, loo, con_u=fu, con_p=fp, lin=l_b)
        np.add.at(ref_ru, cu, ru)
        np.add.at(ref_rp, maps[k - 1], rp)
        lin[cellpos] = l_b.reshape(len(cellpos), -1)
    ref_ru += usr_u
    ref_rp += usr_p
    op = u.engine()
    rhs = op.block_vector()
    op.residual(rhs, op.block_vector(sol_u, sol_p), op.block_vector(usr_u, usr_p), op.block_vector(old_u), op.block_vector(oldold_u))
    got_ru, got_rp = rhs.numpy()
    assert rel_l2(got_ru, ref_ru) < TOL and rel_l2(got_rp, ref_rp) < TOL, (rel_l2(got_ru, ref_ru), rel_l2(got_rp, ref_rp))
    assert rel_l2(op.get_linearization(), lin.reshape(-1)) < TOL
    src_u, src_p = rng.uniform(-1, 1, u.n_u), rng.uniform(-1, 1, u.n_p)
    ref_u, ref_p = u.vmult(src_u, src_p, lin.reshape(-1))
    dst = op.block_vector()
    op.vmult(dst, op.block_vector(src_u, src_p))
    got_u, got_p = dst.numpy()
    assert rel_l2(got_u, ref_u) < TOL and rel_l2(got_p, ref_p) < TOL, (rel_l2(got_u, ref_u), rel_l2(got_p, ref_p))
    # velocity block = vmult of (src_u, 0) restricted to the velocity rows
    ref_v, _ = u.vmult(src_u, np.zeros(u.n_p), lin.reshape(-1), project=False)
    op.fix_linearization_point()
    op.set_linearization(rng.uniform(-1, 1, lin.size))
    vdst = op.initialize_u_vector(np.full(u.n_u, 3.0))
    op.velocity_vmult(vdst, op.initialize_u_vector(src_u))
    assert rel_l2(vdst.numpy(), ref_v) < TOL, rel_l2(vdst.numpy(), ref_v)


def test_scalar_sub_blocks_and_refusals_on_the_union():
    k = 2
    u = Union(k, steps=3, viscosity=0.37)
    rng = np.random.default_rng(5)
    src_u, src_p, base = rng.uniform(-1, 1, u.n_u), rng.uniform(-1, 1, u.n_p), rng.uniform(-1, 1, u.n_p)
    op = u.engine()
    refs = {"div": base.copy(), "poisson": np.zeros(u.n_p), "mass": np.zeros(u.n_p)}
    for part in u.parts:
        omesh, maps, _ = part
        lu, lp = u.local(part, src_u, src_p)
        cu = np.repeat(maps[k], 3) * 3 + np.tile(np.arange(3), len(maps[k]))
        fu, fp = u.flag_u[cu].astype(np.uint8), u.flag_p[maps[k - 1]].astype(np.uint8)
        np.add.at(refs["div"], maps[k - 1], orc.ns_divergence_vmult_add(omesh, k, u.case.prm, lu, np.zeros(len(lp)), fu, fp, mu=None,
                                                                      weight_by_viscosity=True))
        np.add.at(refs["poisson"], maps[k - 1], orc.ns_pressure_poisson_vmult(omesh, k, u.case.prm, lp, fp))
        np.add.at(refs["mass"], maps[k - 1], orc.ns_pressure_mass_vmult(omesh, k, u.case.prm, lp, fp))
    dp = op.initialize_p_vector(base)
    op.divergence_vmult_add(dp, op.initialize_u_vector(src_u), True)
    assert rel_l2(dp.numpy(), refs["div"]) < TOL
    for name in ("poisson", "mass"):
        dp = op.initialize_p_vector(np.full(u.n_p, 5.0))
        getattr(op, "pressure_%s_vmult" % name)(dp, op.initialize_p_vector(src_p))
        assert rel_l2(dp.numpy(), refs[name]) < TOL, name
    # what needs the brick says so
    from adaflo_amd import _lib
    with pytest.raises(_lib.AdafloError, match="generic kernels"):
        op.set_kernel_variant(1)
    d = op.initialize_u_vector(np.zeros(u.n_u))
    with pytest.raises(_lib.AdafloError, match="structured brick"):
        op.velocity_block_diagonal(d)


def test_tables_that_cannot_work_are_refused():
    """two cells of one colour sharing a node (the scatter has no atomics), an index out of range"""
    import ctypes as C
    from adaflo_amd import _lib
    u = Union(2)
    op = adaflo_amd.NavierStokesMatrix(u.case.fp, u.mesh)
    good = u.mesh.colour_offsets.copy()
    u.mesh.colour_offsets = np.array([0, u.mesh.n_cells], dtype=np.int64)      # one colour for all cells
    with pytest.raises(_lib.AdafloError, match="share a"):
        op.initialize(u.case.ts, True)
    u.mesh.colour_offsets = good
    u.mesh.cell_nodes[2][3, 5] = u.mesh.n_nodes(2)
    with pytest.raises(_lib.AdafloError, match="out of range"):
        op.initialize(u.case.ts, True)

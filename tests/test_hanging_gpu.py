"""Hanging nodes on the indexed context (adaflo_indexed_desc.hanging_*): a lattice with cells refined once, as the
reference's Beltrami driver produces (tests/beltrami.cc:403-412), with the constraints of
DoFTools::make_hanging_node_constraints (source/navier_stokes.cc:241-242) -- against the oracle.  The oracle knows bricks
only; with E_c the matrix that reads the nodes of cell c out of a global vector (hanging rows = the constraint weights),
MatrixFree's loop is sum_c E_c^T A_c E_c (FEEvaluation::read_dof_values / distribute_local_to_global,
source/navier_stokes_matrix.cc:232-245): one oracle call per cell on a one-cell brick."""
import numpy as np
import pytest
import scipy.sparse as sp

import adaflo_amd
from common import Case, rel_l2
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-12

NCELL, H = (3, 2, 2), (0.5, 0.4, 0.3)
REFINED = [(1, 0, 0), (2, 1, 1)]                 # two cells, as in beltrami.cc; (2,1,1) touches the boundary


class Refined:
    def __init__(self, k, **kw):
        self.k = k
        self.mesh = m = adaflo_amd.RefinedMesh(NCELL, H, REFINED, k)
        self.case = Case((1, 1, 1), k=k, **kw)
        self.nq = (k + 1) ** 3
        self.n_u, self.n_p = 3 * m.n_nodes(k), m.n_nodes(k - 1)
        self.flag_u, self.flag_p = m.constrained_u.astype(bool), m.constrained_p.astype(bool)
        self.cells = []
        for c in range(m.n_cells):
            omesh = orc.Mesh.make([1, 1, 1], list(m.cell_lower[c]), list(m.cell_lower[c] + m.cell_extents[c]))
            E, loc_flag = {}, {}
            for degree, nc, n_nodes, flags in ((k, 3, m.n_nodes(k), self.flag_u), (k - 1, 1, m.n_nodes(k - 1), self.flag_p)):
                ptr, master, weight = m.hanging[degree]
                rows, cols, vals, lf = [], [], [], []
                for l, node in enumerate(m.cell_nodes[degree][c]):
                    ent = [(int(node), 1.0)] if node >= 0 else [(int(master[j]), float(weight[j])) for j in range(ptr[-1 - node], ptr[-node])]
                    for comp in range(nc):
                        for n, w in ent:
                            rows.append(l * nc + comp), cols.append(n * nc + comp), vals.append(w)
                        lf.append(bool(flags[node * nc + comp]) if node >= 0 else False)
                n_loc = len(m.cell_nodes[degree][c]) * nc
                E[degree] = sp.csr_matrix((vals, (rows, cols)), shape=(n_loc, n_nodes * nc))
                loc_flag[degree] = np.array(lf, dtype=np.uint8)
            self.cells.append((omesh, E, loc_flag))

    def engine(self, pressure_average_fix=True):
        op = adaflo_amd.NavierStokesMatrix(self.case.fp, self.mesh)
        op.initialize(self.case.ts, pressure_average_fix)
        return op

    def q(self, arr, c, width):
        return None if arr is None else arr.reshape(self.mesh.n_cells, width)[c].copy()

    def weights(self):
        w = np.zeros(self.n_p)
        for omesh, E, lf in self.cells:
            w += E[self.k - 1].T @ orc.ns_pressure_mass_weight(omesh, self.k, lf[self.k - 1])
        w[self.flag_p] = 0.
        return w

    def vmult(self, src_u, src_p, lin, coef=(None, None, None), project=True):
        k = self.k
        hom_u, hom_p = np.where(self.flag_u, 0., src_u), np.where(self.flag_p, 0., src_p)   # constrained masters read as zero
        ref_u, ref_p = np.zeros(self.n_u), np.zeros(self.n_p)
        for c, (omesh, E, lf) in enumerate(self.cells):
            du, dp = orc.ns_vmult(omesh, k, self.case.prm, E[k] @ hom_u, E[k - 1] @ hom_p, lf[k], lf[k - 1],
                                  lin=self.q(lin, c, 12 * self.nq), rho=self.q(coef[0], c, self.nq), mu=self.q(coef[1], c, self.nq),
                                  damp=self.q(coef[2], c, self.nq))
            du[lf[k].astype(bool)], dp[lf[k - 1].astype(bool)] = 0., 0.
            ref_u += E[k].T @ du
            ref_p += E[k - 1].T @ dp
        ref_u[self.flag_u], ref_p[self.flag_p] = src_u[self.flag_u], -src_p[self.flag_p]
        if project:
            ref_p = orc.ns_pressure_projection(ref_p, self.weights(), np.where(self.flag_p, 0., 1.))
        return ref_u, ref_p


@pytest.mark.parametrize("k", [2, 3, 4])
@pytest.mark.parametrize("lin_scheme,two_phase", [(0, False), (1, False), (0, True)])
def test_vmult_with_hanging_nodes(k, lin_scheme, two_phase):
    if k == 4 and (lin_scheme, two_phase) != (0, False):
        pytest.skip("k = 4: Newton only (the other branches are covered at k = 2, 3)")
    r = Refined(k, linearization=lin_scheme, tau_grad_div=0.1, damping=0.2, density_diff=0.5 if two_phase else 0.0, steps=3)
    rng = np.random.default_rng(11 + k)
    src_u, src_p = rng.uniform(-1, 1, r.n_u), rng.uniform(-1, 1, r.n_p)
    lin = rng.uniform(-1, 1, r.mesh.n_cells * r.nq * 12)
    coef = tuple(rng.uniform(lo, hi, r.mesh.n_cells * r.nq) for lo, hi in ((.5, 2.), (.5, 2.), (-.5, .5))) if two_phase else (None,) * 3
    ref_u, ref_p = r.vmult(src_u, src_p, lin, coef)
    op = r.engine()
    op.set_linearization(lin)
    if two_phase:
        op.set_coefficients(*coef)
    dst = op.block_vector(np.full(r.n_u, 7.0), np.full(r.n_p, 7.0))
    op.vmult(dst, op.block_vector(src_u, src_p))
    got_u, got_p = dst.numpy()
    assert rel_l2(got_u, ref_u) < TOL and rel_l2(got_p, ref_p) < TOL, (rel_l2(got_u, ref_u), rel_l2(got_p, ref_p))
    hu = np.repeat(r.mesh.hanging_nodes(k), 3) * 3 + np.tile(np.arange(3), len(r.mesh.hanging_nodes(k)))
    assert np.array_equal(got_u[hu], src_u[hu])                           # hanging rows: the identity (:247-256)
    # bitwise reproducible (no atomics in the scatter to the masters)
    dst2 = op.block_vector()
    op.vmult(dst2, op.block_vector(src_u, src_p))
    assert all(np.array_equal(a, b) for a, b in zip(dst2.numpy(), (got_u, got_p)))


@pytest.mark.parametrize("k", [2, 3])
def test_residual_and_sub_blocks_with_hanging_nodes(k):
    r = Refined(k, tau_grad_div=0.1, steps=3, viscosity=0.37)
    rng = np.random.default_rng(4)
    X = r.mesh.node_coordinates(k)
    sol_u = (0.3 * np.stack([np.sin(2 * X[:, 0] + X[:, 1]), np.cos(X[:, 1] - X[:, 2]), np.sin(X[:, 2] + 3 * X[:, 0])], axis=1)
             + 0.05 * rng.uniform(-1, 1, (len(X), 3))).reshape(-1)
    sol_p, old_u, oldold_u = rng.uniform(-1, 1, r.n_p), rng.uniform(-1, 1, r.n_u), rng.uniform(-1, 1, r.n_u)
    usr_u, usr_p = rng.uniform(-1, 1, r.n_u), rng.uniform(-1, 1, r.n_p)
    ref_ru, ref_rp, lin = np.zeros(r.n_u), np.zeros(r.n_p), np.zeros((r.mesh.n_cells, r.nq * 12))
    for c, (omesh, E, lf) in enumerate(r.cells):
        l_b = np.zeros(r.nq * 12)
        ru, rp = orc.ns_residual(omesh, k, r.case.prm, E[k] @ sol_u, E[k - 1] @ sol_p, E[k] @ old_u, E[k] @ oldold_u,
                                 con_u=lf[k], con_p=lf[k - 1], lin=l_b)
        ref_ru += E[k].T @ ru
        ref_rp += E[k - 1].T @ rp
        lin[c] = l_b
    ref_ru[r.flag_u], ref_rp[r.flag_p] = 0., 0.
    ref_ru += usr_u
    ref_rp += usr_p
    op = r.engine()
    rhs = op.block_vector()
    op.residual(rhs, op.block_vector(sol_u, sol_p), op.block_vector(usr_u, usr_p), op.block_vector(old_u), op.block_vector(oldold_u))
    got_ru, got_rp = rhs.numpy()
    assert rel_l2(got_ru, ref_ru) < TOL and rel_l2(got_rp, ref_rp) < TOL, (rel_l2(got_ru, ref_ru), rel_l2(got_rp, ref_rp))
    assert rel_l2(op.get_linearization(), lin.reshape(-1)) < TOL
    # vmult on the residual's state, velocity_vmult on the frozen copy
    src_u, src_p = rng.uniform(-1, 1, r.n_u), rng.uniform(-1, 1, r.n_p)
    ref_u, ref_p = r.vmult(src_u, src_p, lin.reshape(-1))
    dst = op.block_vector()
    op.vmult(dst, op.block_vector(src_u, src_p))
    got_u, got_p = dst.numpy()
    assert rel_l2(got_u, ref_u) < TOL and rel_l2(got_p, ref_p) < TOL
    ref_v, _ = r.vmult(src_u, np.zeros(r.n_p), lin.reshape(-1), project=False)
    op.fix_linearization_point()
    vdst = op.initialize_u_vector(np.full(r.n_u, 3.0))
    op.velocity_vmult(vdst, op.initialize_u_vector(src_u))
    assert rel_l2(vdst.numpy(), ref_v) < TOL
    # scalar sub-blocks
    base = rng.uniform(-1, 1, r.n_p)
    refs = {"div": np.zeros(r.n_p), "poisson": np.zeros(r.n_p), "mass": np.zeros(r.n_p)}
    hom_u, hom_p = np.where(r.flag_u, 0., src_u), np.where(r.flag_p, 0., src_p)
    for omesh, E, lf in r.cells:
        lp = E[k - 1] @ hom_p
        refs["div"] += E[k - 1].T @ orc.ns_divergence_vmult_add(omesh, k, r.case.prm, E[k] @ hom_u, np.zeros(len(lp)), lf[k], lf[k - 1],
                                                                mu=None, weight_by_viscosity=True)
        refs["poisson"] += E[k - 1].T @ orc.ns_pressure_poisson_vmult(omesh, k, r.case.prm, lp, lf[k - 1])
        refs["mass"] += E[k - 1].T @ orc.ns_pressure_mass_vmult(omesh, k, r.case.prm, lp, lf[k - 1])
    free = ~r.flag_p                                                       # (what the constrained rows carry is the caller's)
    dp = op.initialize_p_vector(base)
    op.divergence_vmult_add(dp, op.initialize_u_vector(src_u), True)
    assert rel_l2(dp.numpy()[free], (base + refs["div"])[free]) < TOL
    for name in ("poisson", "mass"):
        dp = op.initialize_p_vector(np.full(r.n_p, 5.0))
        getattr(op, "pressure_%s_vmult" % name)(dp, op.initialize_p_vector(src_p))
        assert rel_l2(dp.numpy()[free], refs[name][free]) < TOL, name


def test_hanging_tables_that_cannot_work_are_refused():
    from adaflo_amd import _lib
    r = Refined(2)
    op = adaflo_amd.NavierStokesMatrix(r.case.fp, r.mesh)
    ptr, master, weight = r.mesh.hanging[2]
    r.mesh.hanging[2] = (ptr, np.where(np.arange(len(master)) == 5, r.mesh.n_nodes(2), master).astype(np.int32), weight)
    with pytest.raises(_lib.AdafloError, match="hanging-node rows"):
        op.initialize(r.case.ts, True)
    r.mesh.hanging[2] = (ptr, master, weight)
    good = r.mesh.colour_offsets.copy()
    # cells that only meet through a MASTER of a hanging node must not share a colour: put all cells in one colour
    r.mesh.colour_offsets = np.array([0, r.mesh.n_cells], dtype=np.int64)
    with pytest.raises(_lib.AdafloError, match="share a"):
        op.initialize(r.case.ts, True)
    r.mesh.colour_offsets = good
    op.initialize(r.case.ts, True)

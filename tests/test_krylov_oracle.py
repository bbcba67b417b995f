"""CPU tier: the numpy Krylov restatement (oracle/krylov_oracle.py) on the oracle's level-set
operators: it solves the systems the reference solves with SolverCG / SolverBicgstab, and the
solutions satisfy the equations to the requested reduction."""
import numpy as np
import pytest

from oracle import krylov_oracle as ko
from oracle import oracle as orc


def ls_setup(ncell=(3, 3, 4), s=2):
    mesh = orc.Mesh.make(list(ncell), (0., 0., 0.), (1., 1., 2.))
    h = [mesh.h[d] for d in range(3)]
    prm = orc.make_ls_params(s, 1.5 * max(h) / s, min(h), 0.02, 75.0, max(h), 1.5)
    return mesh, prm, mesh.n_nodes(s), (2 * s) ** 3


def probe_diagonal(A, n):
    d = np.empty(n)
    e = np.zeros(n)
    for i in range(n):
        e[i] = 1.0
        d[i] = A(e)[i]
        e[i] = 0.0
    return d


def test_cg_solves_the_curvature_projection_to_the_requested_reduction():
    mesh, prm, nn, _ = ls_setup((2, 2, 3), 2)
    A = lambda v: orc.ls_curvature_vmult(mesh, prm, v)
    rng = np.random.default_rng(0)
    b = rng.uniform(-1, 1, nn)
    inv = 1.0 / probe_diagonal(A, nn)
    x, its, r0, r, ok = ko.cg(A, b, inv_diag=inv, rel_tol=1e-8)       # compute_curvature.cc:347
    assert ok and 0 < its < 60
    assert np.linalg.norm(b - A(x)) <= 1.01e-8 * np.linalg.norm(b)
    # the preconditioner pays off on this mass-dominated operator
    _, its_plain, *_ = ko.cg(A, b, rel_tol=1e-8)
    assert its <= its_plain


def test_bicgstab_solves_the_advection_system():
    mesh, prm, nn, nq = ls_setup((3, 3, 3), 2)
    rng = np.random.default_rng(1)
    uq = rng.uniform(-0.3, 0.3, mesh.n_cells * nq * 3)
    A = lambda v: orc.ls_advect_vmult(mesh, prm, v, uq)
    b = rng.uniform(-1, 1, nn)
    inv = 1.0 / probe_diagonal(A, nn)
    x, its, r0, r, ok = ko.bicgstab(A, b, inv_diag=inv, max_it=200, rel_tol=1e-8)   # advance_concentration.cc:629
    assert ok and its < 200
    assert np.linalg.norm(b - A(x)) <= 1e-7 * np.linalg.norm(b)


def test_cg_reports_failure_when_the_step_limit_is_hit():
    mesh, prm, nn, _ = ls_setup((2, 2, 2), 2)
    A = lambda v: orc.ls_curvature_vmult(mesh, prm, v)
    b = np.random.default_rng(2).uniform(-1, 1, nn)
    *_, ok = ko.cg(A, b, max_it=2, rel_tol=1e-12)
    assert not ok

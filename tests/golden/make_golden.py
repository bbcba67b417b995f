"""Generate the committed golden fixtures under tests/golden/.

Two kinds of data live there:
  * reference_outputs.json -- numbers TRANSCRIBED BY HAND from the reference's own test outputs
    (tests/beltrami_3d.output, tests/rising_bubble_ls{,_picard,_imex,_expl,_q3}.output,
    tests/spurious_currents_ls.output): DoF counts, iteration counts and residual norms as printed.
    They pin the oracle (tests/test_oracle_golden*.py).  Not written by this script.
  * *.npz -- seeded inputs and the outputs of the CPU oracle (oracle/adaflo_oracle.c) for small
    meshes, every operator on the path: written by this script.  They freeze the oracle against
    drift and let the HIP engine be checked against data that does not depend on building the
    oracle at test time.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def ns_case(name, ncell, k, lower, upper, linearization=0, physical_type=0, variable=False, seed=1):
    dim = len(ncell)
    rng = np.random.default_rng(seed)
    mesh = orc.Mesh.make(list(ncell), lower, upper)
    prm = orc.NSParams.make(physical_type=physical_type, linearization=linearization, beta=0.5,
                            tau_grad_div=0.1,
                            density=0.0 if physical_type == 2 else 1.1,  # parameters.cc:477-478
                            viscosity=0.7, damping=-0.2,
                            density_diff=0.3 if variable else 0.0, weight=30.0, weight_old=-40.0,
                            weight_old_old=10.0, tau1=1.0, extrap_old=2.0, extrap_old_old=-1.0)
    n_u, n_p = mesh.n_nodes(k) * dim, mesh.n_nodes(k - 1)
    nq = (k + 1) ** dim
    con_u = orc.boundary_mask(mesh, k, dim)
    con_p = orc.boundary_mask(mesh, k - 1, 1, faces=[0])
    d = dict(ncell=np.array(ncell), k=k, lower=np.array(lower), upper=np.array(upper),
             linearization=linearization, physical_type=physical_type,
             prm=np.array([getattr(prm, f) for f, _ in orc.NSParams._fields_], dtype=float),
             src_u=rng.uniform(-1, 1, n_u), src_p=rng.uniform(-1, 1, n_p),
             old_u=rng.uniform(-1, 1, n_u), oldold_u=rng.uniform(-1, 1, n_u),
             lin=rng.uniform(-1, 1, mesh.n_cells * nq * orc.n_lin(dim)))
    co = {}
    if variable:
        n = mesh.n_cells * nq
        d["rho"], d["mu"], d["damp"] = rng.uniform(.5, 2, n), rng.uniform(.5, 2, n), rng.uniform(-.5, .5, n)
        co = dict(rho=d["rho"], mu=d["mu"], damp=d["damp"])
    d["vmult_u"], d["vmult_p"] = orc.ns_vmult(mesh, k, prm, d["src_u"], d["src_p"], con_u, con_p, lin=d["lin"], **co)
    d["velocity_vmult"] = orc.ns_velocity_vmult(mesh, k, prm, d["src_u"], con_u, lin=d["lin"], **co)
    lin_out = np.zeros_like(d["lin"])
    d["residual_u"], d["residual_p"] = orc.ns_residual(mesh, k, prm, d["src_u"], d["src_p"], d["old_u"],
                                                       d["oldold_u"], con_u=con_u, con_p=con_p, lin=lin_out, **co)
    d["residual_lin"] = lin_out
    d["divergence_add"] = orc.ns_divergence_vmult_add(mesh, k, prm, d["src_u"], d["src_p"], con_u, con_p)
    if prm.density > 0:  # Stokes: density = 0 and the operator (coefficient 1/(weight rho)) is not used
        d["pressure_poisson"] = orc.ns_pressure_poisson_vmult(mesh, k, prm, d["src_p"], con_p, rho=co.get("rho"))
    d["pressure_mass"] = orc.ns_pressure_mass_vmult(mesh, k, prm, d["src_p"], con_p, mu=co.get("mu"))
    d["pressure_mass_weight"] = orc.ns_pressure_mass_weight(mesh, k, con_p)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    return name


def ls_case(name, ncell, s, k=2, seed=2):
    rng = np.random.default_rng(seed)
    lower, upper = (0., 0., 0.), (1., 1., 2.)
    mesh = orc.Mesh.make(list(ncell), lower, upper)
    h = [mesh.h[d] for d in range(3)]
    eps_used, dt, weight, w_old, w_oo, epsilon = 1.5 * max(h) / s, 0.02, 75.0, -100.0, 25.0, 1.5
    prm = orc.make_ls_params(s, eps_used, min(h), dt, weight, max(h), epsilon)
    nn, nq = mesh.n_nodes(s), (2 * s) ** 3
    con = orc.boundary_mask(mesh, s, 1, faces=[0, 5])
    d = dict(ncell=np.array(ncell), s=s, k=k, lower=np.array(lower), upper=np.array(upper),
             scalars=np.array([eps_used, dt, weight, w_old, w_oo, epsilon]),
             src=rng.uniform(-1, 1, nn), src3=rng.uniform(-1, 1, 3 * nn), diag=rng.uniform(.5, 2, nn),
             vel_q=rng.uniform(-1, 1, mesh.n_cells * nq * 3), normal_q=rng.uniform(-1, 1, mesh.n_cells * nq * 3),
             old=rng.uniform(-1, 1, nn), oldold=rng.uniform(-1, 1, nn),
             vel=rng.uniform(-1, 1, mesh.n_nodes(k) * 3))
    d["advect_vmult"] = orc.ls_advect_vmult(mesh, prm, d["src"], d["vel_q"], con=con, diag=d["diag"])
    d["reinit_vmult"] = orc.ls_reinit_vmult(mesh, prm, d["src"], d["normal_q"], con=con, diag=d["diag"])
    d["reinit_diffuse_vmult"] = orc.ls_reinit_vmult(mesh, prm, d["src"], d["normal_q"], diffuse_only=True,
                                                    con=con, diag=d["diag"])
    d["normal_vmult"] = orc.ls_normal_vmult(mesh, prm, d["src3"], con=con, diag=d["diag"])
    d["curvature_vmult"] = orc.ls_curvature_vmult(mesh, prm, d["src"], con=con, diag=d["diag"])
    # right-hand sides are formed without constraints in the parity tests
    nq_out = np.zeros(mesh.n_cells * nq * 3)
    d["reinit_rhs_first"] = orc.ls_reinit_rhs(mesh, prm, d["src"], d["src3"], nq_out, diffuse_only=False, first_step=True)
    d["reinit_rhs_normal_q"] = nq_out
    d["normal_rhs"] = orc.ls_normal_rhs(mesh, prm, d["src"])
    d["curvature_rhs"] = orc.ls_curvature_rhs(mesh, prm, d["src3"])
    uq = np.zeros(mesh.n_cells * nq * 3)
    d["advect_rhs"] = orc.ls_advect_rhs(mesh, prm, k, d["src"], d["old"], d["oldold"], d["vel"], uq, w_old, w_oo, True)
    d["advect_rhs_vel_q"] = uq
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    return name


def main():
    orc.build()
    os.makedirs(OUT, exist_ok=True)
    names = [
        ns_case("ns_2d_q2_8x8_newton", (8, 8), 2, (-1., -1.), (1., 1.)),
        ns_case("ns_3d_q2_4x4x4_newton", (4, 4, 4), 2, (-1., -1., -1.), (1., 1., 1.)),
        ns_case("ns_3d_q2_9x8x3_newton", (9, 8, 3), 2, (-1., -1., -1.), (1., 0.5, 2.)),
        ns_case("ns_3d_q2_4x3x4_picard_variable", (4, 3, 4), 2, (0., 0., 0.), (1., 1., 3.), linearization=1, variable=True),
        ns_case("ns_3d_q2_4x4x3_semi_implicit", (4, 4, 3), 2, (0., 0., 0.), (1., 1., 1.), linearization=2),
        ns_case("ns_3d_q2_3x3x3_stokes", (3, 3, 3), 2, (0., 0., 0.), (1., 1., 1.), physical_type=2),
        ns_case("ns_3d_q3_3x3x3_newton", (3, 3, 3), 3, (-1., -1., -1.), (1., 1., 1.)),
        ns_case("ns_3d_q4_3x2x2_newton", (3, 2, 2), 4, (0., 0., 0.), (1., 1., 3.)),
        ls_case("ls_3d_s4_2x2x3", (2, 2, 3), 4),
        ls_case("ls_3d_s2_5x9x3", (5, 9, 3), 2),
    ]
    for n in names:
        print(n, os.path.getsize(os.path.join(OUT, n + ".npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()

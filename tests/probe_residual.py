"""TEST INFRASTRUCTURE (development probe, not collected by pytest): repeat NavierStokesMatrix::residual of one Q_k/Q_{k-1}
case on the sweep / x-marching kernels and print the errors against the oracle, per z-plane and component.
   usage (repo root, GPU box): python tests/probe_residual.py k,nx,ny,nz,linearization [...]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from common import Case, rel_l2
from oracle import oracle as orc

def run(k, ncell, lin, reps=4):
    case = Case(ncell, k=k, lower=(0., 0., 0.), upper=(1., 1.5, 1.), faces_u=[0, 2, 3, 5], faces_p=[1],
                linearization=lin, physical_type=0, tau_grad_div=0.2, damping=0.1, density=1.2, steps=3)
    src_u, src_p = case.smooth_u(0.1) + 0.01 * case.random_u(), case.smooth_p(0.1)
    old_u, oldold_u = case.smooth_u(0.05), case.smooth_u(0.0)
    lin_ref = np.zeros(case.n_cells * case.nq * 12)
    ref_u, ref_p = orc.ns_residual(case.mesh, k, case.prm, src_u, src_p, old_u, oldold_u, con_u=case.con_u, con_p=case.con_p, lin=lin_ref)
    for r in range(reps):
        op = case.engine()
        op.set_kernel_variant(1)
        rhs = op.block_vector()
        op.residual(rhs, op.block_vector(src_u, src_p), None, op.block_vector(old_u), op.block_vector(oldold_u))
        gu, gp = rhs.numpy()
        bad = np.nonzero(np.abs(gp - ref_p) > 1e-9 * np.abs(ref_p).max())[0]
        if len(bad) and r == 0:
            for b_ in bad[:8]:
                print("   row", b_, "got", gp[b_], "ref", ref_p[b_], "diff", gp[b_] - ref_p[b_])
        if r == 0:
            nnx, nny, nnz = [k * n + 1 for n in ncell]
            eu = np.abs(gu - ref_u).reshape(nnz, nny, nnx, 3)
            print("   max |err_u| per z-plane:", " ".join("%.1e" % v for v in eu.max(axis=(1, 2, 3))))
            print("   max |err_u| per component:", " ".join("%.1e" % v for v in eu.max(axis=(0, 1, 2))), " per x:", " ".join("%.0e" % v for v in eu.max(axis=(0, 1, 3))))
        print(k, ncell, "lin", lin, "rep", r, "err_u %.2e err_p %.2e" % (rel_l2(gu, ref_u), rel_l2(gp, ref_p)), "bad p rows", bad[:12], len(bad), flush=True)

for a in sys.argv[1:]:
    k, nx, ny, nz, lin = map(int, a.split(","))
    run(k, (nx, ny, nz), lin)

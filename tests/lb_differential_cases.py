"""TEST INFRASTRUCTURE (run by tests/test_lb_differential_gpu.py, not collected by pytest): outputs of the kernels that ship
at one workgroup per CU (512 registers: residual modes of the Q2/Q1 and x-marching kernels, recompute mode on non-cubic
cells, the extrapolating residuals) for a set of meshes, written to <out>.npz -- run once with the product library and
once with a library built for 256 registers (adaflo_amd/build.py: VARIANTS) and compared bitwise: the same source, the same
floating-point operations, another register allocation.
usage: ADAFLO_LIB_PATH=... python tests/lb_differential_cases.py out.npz"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import adaflo_amd

out = {}
rng = np.random.default_rng(7)
for k, cells, upper, lin in ((2, (24, 17, 12), (1., 1., 1.), "coupled implicit Newton"), (2, (33, 16, 9), (1., 2., 1.), "coupled implicit Newton"),
                             (2, (16, 16, 16), (1., 1., 1.), "coupled implicit Picard"), (2, (24, 17, 12), (1., 1., 1.), "coupled velocity semi-implicit"),
                             (4, (9, 7, 10), (1., 1., 1.), "coupled implicit Newton"), (5, (5, 3, 6), (1., 1., 1.), "coupled implicit Newton"),
                             (5, (4, 4, 4), (1., 1., 1.), "coupled implicit Picard"), (4, (8, 8, 8), (1., 1., 1.), "coupled velocity semi-implicit"),
                             (3, (9, 9, 9), (1., 1., 1.), "coupled velocity explicit"), (2, (8, 8, 4), (1., 1., 1.), "coupled velocity explicit"),
                             (2, (17, 9, 6), (1., 1.5, 1.), "coupled velocity explicit"), (5, (3, 2, 3), (1., 1.5, 1.), "coupled velocity semi-implicit"),
                             (5, (4, 4, 4), (1., 1., 1.), "coupled velocity explicit"), (4, (5, 3, 4), (1., 1., 1.), "coupled velocity explicit")):
    # (two-phase variants since round 6 -- the variable-coefficient residuals of every scheme are new 512-register builds)
    for two_phase in ((False, True) if k in (2, 4, 5) else (False,)):
        fp = adaflo_amd.FlowParameters(velocity_degree=k, linearization=lin, density_diff=0.5 if two_phase else 0.0)
        ts = adaflo_amd.TimeStepping(fp)
        for _ in range(3):
            ts.next()
        op = adaflo_amd.NavierStokesMatrix(fp, adaflo_amd.BrickMesh(list(cells), [0.] * 3, list(upper)))
        op.initialize(ts, True)
        if two_phase:
            nc = op.n_cells() * (k + 1) ** 3
            op.set_coefficients(rng.uniform(.5, 2, nc), rng.uniform(.5, 2, nc), rng.uniform(-.5, .5, nc))
        sol = op.block_vector(rng.uniform(-1, 1, op.n_dofs_u()), rng.uniform(-1, 1, op.n_dofs_p()))
        old = adaflo_amd.BlockVector([op.initialize_u_vector(rng.uniform(-1, 1, op.n_dofs_u()))])
        oldold = adaflo_amd.BlockVector([op.initialize_u_vector(rng.uniform(-1, 1, op.n_dofs_u()))])
        rhs, dst = op.block_vector(), op.block_vector()
        op.residual(rhs, sol, None, old, oldold)
        key = "k%d_%s_%s_%s" % (k, "x".join(map(str, cells)), lin.replace(" ", "_"), "tp" if two_phase else "cc")
        ru, rp = rhs.numpy()
        out[key + "_res_u"], out[key + "_res_p"] = ru, rp
        if "explicit" not in lin:
            out[key + "_lin"] = op.get_linearization()
        op.vmult(dst, sol)
        vu, vp = dst.numpy()
        out[key + "_vm_u"], out[key + "_vm_p"] = vu, vp
np.savez(sys.argv[1], **out)
print("wrote", sys.argv[1], len(out), "arrays")

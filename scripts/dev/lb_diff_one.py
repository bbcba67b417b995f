"""development: residual of one Q2/Q1 case with the library given by ADAFLO_LIB_PATH, written to <out>.npy (for bitwise
comparison of two builds of the same source).  usage: lb_diff_one.py out.npy linearization-name nx ny nz"""
import sys
sys.path.insert(0, ".")
import numpy as np
import adaflo_amd
out, lin, nx, ny, nz = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
rng = np.random.default_rng(7)
fp = adaflo_amd.FlowParameters(velocity_degree=2, linearization=lin)
ts = adaflo_amd.TimeStepping(fp)
for _ in range(3):
    ts.next()
op = adaflo_amd.NavierStokesMatrix(fp, adaflo_amd.BrickMesh([nx, ny, nz], [0.] * 3, [1.] * 3))
op.initialize(ts, True)
sol = op.block_vector(rng.uniform(-1, 1, op.n_dofs_u()), rng.uniform(-1, 1, op.n_dofs_p()))
old = adaflo_amd.BlockVector([op.initialize_u_vector(rng.uniform(-1, 1, op.n_dofs_u()))])
oldold = adaflo_amd.BlockVector([op.initialize_u_vector(rng.uniform(-1, 1, op.n_dofs_u()))])
rhs = op.block_vector()
op.residual(rhs, sol, None, old, oldold)
np.save(out, np.concatenate(rhs.numpy()))

"""development: residual of the explicit scheme for u_0 = x_0, p = 0, old solutions zero, written to <out>.npy (see
scripts/dev/isa_patch_build.py, edit `poison`)"""
import sys
sys.path.insert(0, ".")
import numpy as np
import adaflo_amd
fp = adaflo_amd.FlowParameters(velocity_degree=2, linearization="coupled velocity explicit")
ts = adaflo_amd.TimeStepping(fp)
for _ in range(3):
    ts.next()
op = adaflo_amd.NavierStokesMatrix(fp, adaflo_amd.BrickMesh([8, 8, 4], [0.] * 3, [1.] * 3))
op.initialize(ts, True)
z, y, x = np.meshgrid(np.linspace(0, 1, 9), np.linspace(0, 1, 17), np.linspace(0, 1, 17), indexing="ij")
res = []
for c, X in ((0, x), (1, y), (2, z)):
    f = np.zeros(x.shape + (3,))
    f[..., c] = X
    sol = op.block_vector(f.reshape(-1), np.zeros(op.n_dofs_p()))
    zero = adaflo_amd.BlockVector([op.initialize_u_vector(np.zeros(op.n_dofs_u()))])
    rhs = op.block_vector()
    op.residual(rhs, sol, None, zero, zero)
    res.append(np.concatenate(rhs.numpy()))
np.save(sys.argv[1], np.stack(res))

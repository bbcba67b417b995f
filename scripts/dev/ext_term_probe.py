"""development (round 6): which term of the extrapolating residual does a build get wrong?  The explicit scheme adds
cB (beta div(u_ext) u_ext + (u_ext . grad) u_ext) at every quadrature point (navier_stokes_matrix.cc:740-782); with the
solution zero and u_old_old = 4 u_old the time-derivative terms cancel (BDF-2: -2 + 0.5 * 4) and u_ext = -2 u_old remains.
Fields with ONE constant component c and ONE component d linear in x_e isolate the product  u_ext,c * d_e u_ext,d  (the
broadcast of lane c's value times lane d's own derivative).  Run once per library, compare the two result files:
   ADAFLO_LIB_PATH=... python scripts/dev/ext_term_probe.py out.npz ;  python scripts/dev/ext_term_probe.py --compare a.npz b.npz"""
import sys
sys.path.insert(0, ".")
import numpy as np

if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for key in a.files:
        x, y = a[key], b[key]
        nu = 17 * 17 * 9 * 3
        du = np.abs(x[:nu] - y[:nu]).reshape(9, 17, 17, 3)
        w = np.unravel_index(np.argmax(du), du.shape)
        print("%-44s max |diff| u per row component %s  p %.1e   (|ref| %.2e)  worst at node (k, j, i, comp) %s: %.6e vs %.6e" % (
            key, " ".join("%.1e" % v for v in du.max(axis=(0, 1, 2))), np.abs(x[nu:] - y[nu:]).max(), np.abs(y).max(), w,
            x[:nu].reshape(9, 17, 17, 3)[w], y[:nu].reshape(9, 17, 17, 3)[w]))
    for key in ("solution u[0] = x_0^1", "solution u[1] = x_1^1"):
        c = int(key[11])
        d = (a[key][:17 * 17 * 9 * 3] - b[key][:17 * 17 * 9 * 3]).reshape(9, 17, 17, 3)[..., c]
        np.set_printoptions(linewidth=250, precision=2, suppress=True)
        for kz in (1, 2, 3, 4):
            print(key, "difference in row component", c, "plane k =", kz, "(rows j = 0..16, columns i = 0..16)")
            print(d[kz])
    sys.exit(0)

import adaflo_amd
fp = adaflo_amd.FlowParameters(velocity_degree=2, linearization="coupled velocity explicit")
ts = adaflo_amd.TimeStepping(fp)
for _ in range(3):
    ts.next()
op = adaflo_amd.NavierStokesMatrix(fp, adaflo_amd.BrickMesh([8, 8, 4], [0.] * 3, [1.] * 3))
op.initialize(ts, True)
z, y, x = np.meshgrid(np.linspace(0, 1, 9), np.linspace(0, 1, 17), np.linspace(0, 1, 17), indexing="ij")
X = (x, y, z)
out = {}


def run(name, field, su=None, sp=None, oldold_factor=4.):
    u_old = (-0.5 * field).reshape(-1)                   # u_ext = 2 u_old - u_old_old = -2 u_old = field
    sol = op.block_vector(np.zeros(op.n_dofs_u()) if su is None else su, np.zeros(op.n_dofs_p()) if sp is None else sp)
    old = adaflo_amd.BlockVector([op.initialize_u_vector(u_old)])
    oldold = adaflo_amd.BlockVector([op.initialize_u_vector(oldold_factor * u_old)])
    rhs = op.block_vector()
    op.residual(rhs, sol, None, old, oldold)
    out[name] = np.concatenate(rhs.numpy())


for c in range(3):
    for d in range(3):
        for e in range(3):
            f = np.zeros(x.shape + (3,))
            f[..., c] += 1.
            f[..., d] += X[e]
            run("u_ext[%d] = 1, u_ext[%d] += x_%d" % (c, d, e), f)
# which INPUT excites the difference: solution only (velocity / pressure / single components), old solutions only
rng = np.random.default_rng(3)
zero = np.zeros(x.shape + (3,))
ru, rp = rng.uniform(-1, 1, op.n_dofs_u()), rng.uniform(-1, 1, op.n_dofs_p())
run("solution random (u, p), old solutions zero", zero, ru, rp)
run("solution random u, p = 0", zero, ru, None)
run("solution u = 0, random p", zero, None, rp)
for c in range(3):
    only = np.zeros((op.n_dofs_u() // 3, 3))
    only[:, c] = ru.reshape(-1, 3)[:, c]
    run("solution: random component %d only" % c, zero, only.reshape(-1), None)
smooth = np.stack([np.sin(2 * x + y), np.cos(y - z), np.sin(z + 3 * x)], axis=-1)
run("solution smooth u, p = 0", zero, smooth.reshape(-1), None)
run("solution zero, random u_old, u_old_old = 0", rng.uniform(-1, 1, x.shape + (3,)), None, None, 0.)
run("solution zero, random u_old, u_old_old = 4 u_old", rng.uniform(-1, 1, x.shape + (3,)))
run("everything random", rng.uniform(-1, 1, x.shape + (3,)), ru, rp, 0.3)
# polynomial solutions: constant (value terms only), linear and quadratic in one direction (gradient terms)
for c in range(3):
    f = np.zeros(x.shape + (3,))
    f[..., c] = 1.
    run("solution u[%d] = 1" % c, zero, f.reshape(-1), None)
    for e in range(3):
        for pw in (1, 2):
            f = np.zeros(x.shape + (3,))
            f[..., c] = X[e] ** pw
            run("solution u[%d] = x_%d^%d" % (c, e, pw), zero, f.reshape(-1), None)
np.savez(sys.argv[1], **out)

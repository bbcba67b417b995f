import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import Case
from oracle import oracle as orc
ncell = tuple(int(x) for x in sys.argv[1:4])
case = Case(ncell, k=2, upper=(1.0, 0.5, 2.0))
src_u, src_p, lin = case.random_u(), case.random_p(), case.random_lin()
w, modes = case.weights_modes()
ref_u, ref_p = orc.ns_vmult(case.mesh, 2, case.prm, src_u, src_p, case.con_u, case.con_p, lin=lin, weights=w, modes=modes)
op = case.engine(); op.set_linearization(lin)
dst = op.block_vector(); op.vmult(dst, op.block_vector(src_u, src_p))
gu, gp = dst.numpy()
npn = [n + 1 for n in ncell]
err = np.abs(gp - ref_p).reshape(npn[2], npn[1], npn[0])
print("max err", err.max(), "mean shift", (gp - ref_p).mean())
bad = np.argwhere(err > 1e-10)
print("n bad", len(bad), "of", err.size)
print("K values", sorted(set(bad[:, 0]))[:20]); print("J values", sorted(set(bad[:, 1]))[:20]); print("I values", sorted(set(bad[:, 2]))[:20])
# without projection
case2 = Case(ncell, k=2, upper=(1.0, 0.5, 2.0), pressure_average_fix=False)
ref_u2, ref_p2 = orc.ns_vmult(case2.mesh, 2, case2.prm, src_u, src_p, case2.con_u, case2.con_p, lin=lin)
op2 = case2.engine(); op2.set_linearization(lin)
d2 = op2.block_vector(); op2.vmult(d2, op2.block_vector(src_u, src_p))
e2 = np.abs(d2.numpy()[1] - ref_p2).reshape(npn[2], npn[1], npn[0])
bad = np.argwhere(e2 > 1e-10)
print("no-projection: n bad", len(bad)); print(bad[:30])

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
import adaflo_amd
from adaflo_amd import _lib
from adaflo_amd.navier_stokes import NavierStokes, node_coordinates
from oracle import oracle as orc, newton_oracle as no
orc.build()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
fp = adaflo_amd.FlowParameters(velocity_degree=2, viscosity=1.0, time_step_size_start=0.05, end_time=1.0,
                               max_nl_iteration=3, tol_nl_iteration=1e-9, max_lin_iteration=100, tol_lin_iteration=1e-5)
mesh = adaflo_amd.BrickMesh([n] * 3, [-1.0] * 3, [1.0] * 3)
ns = NavierStokes(fp, mesh, adaflo_amd.TimeStepping(fp), dirichlet_function=lambda x, t: orc.beltrami_u(x, t, 1.0).reshape(-1, 3))
xu, xp = node_coordinates(mesh, 2), node_coordinates(mesh, 1)
ns.set_initial_condition(orc.beltrami_u(xu, 0.0, 1.0), orc.beltrami_p(xp, 0.0, 1.0))
st = no.BeltramiStepper(n, adaflo_amd.TimeStepping(fp))
ns.init_time_advance(); st.init_time_advance()
print("u diff", np.abs(ns.solution[0].cpu().numpy() - st.u).max(), "p diff", np.abs(ns.solution[1].cpu().numpy() - st.p).max())
print("old diff", np.abs(ns.solution_old[0].cpu().numpy() - st.u_old).max())
r = ns.compute_residual(); ru, rp = st.residual()
print("res dev", ns.history[-1], "oracle", np.linalg.norm(ru), np.linalg.norm(rp))
print("rhs diff", np.abs(ns.system_rhs[0].cpu().numpy() - ru).max(), np.abs(ns.system_rhs[1].cpu().numpy() - rp).max())
ns.build_preconditioner()
its, lr = ns.solve_system(1e-8)
du, dp = ns.solution_update[0].cpu().numpy(), ns.solution_update[1].cpu().numpy()
A, prm = st.jacobian()
res = np.concatenate([ru, rp]) - A(np.concatenate([du, dp]))
print("fgmres its", its, lr, "true residual with oracle J:", np.linalg.norm(res), "vs rhs", np.linalg.norm(np.concatenate([ru, rp])))
# device vmult of the update
m = ns.navier_stokes_matrix
out = [torch.zeros_like(ns.solution_update[0]), torch.zeros_like(ns.solution_update[1])]
m.vmult(ns._bv(out), ns._bv(ns.solution_update))
print("device A*du vs rhs:", float((out[0] - ns.system_rhs[0]).norm()), float((out[1] - ns.system_rhs[1]).norm()))
# ---- continue: apply the update on both sides, compare residuals
ns.solution[0] += ns.solution_update[0]; ns.solution[1] += ns.solution_update[1]
st.u += du; st.p += dp
r = ns.compute_residual(); ru, rp = st.residual()
print("after update: dev", ns.history[-1], "oracle", np.linalg.norm(ru), np.linalg.norm(rp))
print("state diff after update", np.abs(ns.solution[0].cpu().numpy() - st.u).max(), np.abs(ns.solution[1].cpu().numpy() - st.p).max())
print("rhs diff", np.abs(ns.system_rhs[0].cpu().numpy() - ru).max(), np.abs(ns.system_rhs[1].cpu().numpy() - rp).max())
# fresh engine, same state
from adaflo_amd import NavierStokesMatrix
m2 = NavierStokesMatrix(fp, mesh); m2.initialize(ns.time_stepping, True)
rhs2 = m2.block_vector()
m2.residual(rhs2, m2.block_vector(st.u, st.p), None, m2.block_vector(st.u_old), m2.block_vector(st.u_oldold))
a, b = rhs2.numpy()
print("fresh engine residual", np.linalg.norm(a), np.linalg.norm(b - 0*b))

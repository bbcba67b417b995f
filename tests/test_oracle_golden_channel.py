"""Pin the single-phase Navier-Stokes part of the oracle (2D, open boundaries, symmetry, moving wall) to three more of the
reference's golden outputs: tests/poiseuille_stokes.output, tests/poiseuille_ns.output, tests/couette.output
(oracle/channel_oracle.py: exact Newton steps on the oracle's operators).  tests/test_dim2_gpu.py reproduces the same
lines with the device drivers."""
import json
import os

import numpy as np
from threadpoolctl import threadpool_limits

import adaflo_amd
from oracle import channel_oracle as co
from oracle import oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_outputs.json")


def _ref(case):
    with open(GOLDEN) as f:
        return json.load(f)[case]


def test_poiseuille_navier_stokes_matches_its_reference_output():
    """poiseuille_ns.output:11,31,40,49: first nonlinear residuals of four BDF-2 steps from rest; :56 ||e_u||_L2 = 0.1321"""
    fp = adaflo_amd.FlowParameters(velocity_degree=2, viscosity=0.5, time_step_size_start=0.5, end_time=20.0)
    with threadpool_limits(limits=1, user_api="blas"):
        sim = co.ChannelFlow(adaflo_amd.TimeStepping(fp), ncell=(64, 16), viscosity=0.5)
        assert (sim.mesh.n_cells, sim.nu, sim.np_) == (1024, 8514, 1105)
        for expected in _ref("poiseuille_ns")["first_residuals"]:
            history = sim.advance_time_step()
            assert "%.3e" % history[0] == expected and history[-1] < 1e-11 and len(history) <= 5
    xq, wq = orc.gauss_legendre(4)
    S, _ = orc.shape_1d(0, 2, xq)
    uu = sim.u.reshape(33, 129, 2)
    iy = np.arange(16)[:, None] * 2 + np.arange(3)[None, :]
    ix = np.arange(64)[:, None] * 2 + np.arange(3)[None, :]
    val = np.einsum("qj,pi,yxjic->yxqpc", S, S, uu[iy[:, None, :, None], ix[None, :, None, :]])
    yq = -1.0 + (np.arange(16)[:, None] + xq[None, :]) / 16.0
    exact = np.zeros_like(val)
    exact[..., 0] = ((1 - yq ** 2))[:, None, :, None]                     # 0.5 / nu (1 - y^2)
    err = np.sqrt(np.einsum("yxqpc,qp->", (val - exact) ** 2, np.outer(wq, wq) / 256.0))
    assert "%.4g" % err == _ref("poiseuille_ns")["l2_error_u_after_four_steps"]


def test_couette_matches_its_reference_output():
    """couette.output:10,31"""
    fp = adaflo_amd.FlowParameters(velocity_degree=2, viscosity=0.5, time_step_size_start=0.5, end_time=1.0)
    with threadpool_limits(limits=1, user_api="blas"):
        sim = co.ChannelFlow(adaflo_amd.TimeStepping(fp), ncell=(64, 16), viscosity=0.5, p_ext=lambda x: np.zeros(len(x)),
                             wall_velocity=(2.0, 0.0))
        for expected in _ref("couette")["first_residuals"]:
            history = sim.advance_time_step()
            assert "%.3e" % history[0] == expected and history[-1] < 1e-11


def test_poiseuille_stokes_first_residual_and_exact_solution():
    """poiseuille_stokes.output:11 on 128 x 32 cells would print another number; the reference's mesh (256 x 64) gives
    3.722e-01; one Newton step of the linear problem lands on u = (1 - y^2) / (2 nu), p = 2 - x (representable)"""
    fp = adaflo_amd.FlowParameters(velocity_degree=2, physical_type="stokes", viscosity=0.1, time_step_size_start=0.01, end_time=1.0)
    with threadpool_limits(limits=1, user_api="blas"):
        sim = co.ChannelFlow(adaflo_amd.TimeStepping(fp), ncell=(256, 64), viscosity=0.1, stokes=True)
        assert (sim.mesh.n_cells, sim.nu, sim.np_) == (16384, 132354, 16705)
        sim.ts.next()
        ru, rp = orc.ns_residual(sim.mesh, 2, sim.params(), sim.u, sim.p, sim.u_old, sim.u_oo, con_u=sim.con_u,
                                 lin=np.zeros(16384 * 9 * 6), rhs_u=sim.const_rhs)
        assert "%.3e" % np.hypot(np.linalg.norm(ru), np.linalg.norm(rp)) == _ref("poiseuille_stokes")["first_residual"]
    small = co.ChannelFlow(adaflo_amd.TimeStepping(fp), ncell=(32, 8), viscosity=0.1, stokes=True)
    history = small.advance_time_step(tol_nl=1e-10)
    assert history[-1] < 1e-10 and len(history) == 2
    assert np.abs(small.u.reshape(-1, 2)[:, 0] - 5.0 * (1 - small.x[:, 1] ** 2)).max() < 1e-9
    xp = orc.node_coordinates(small.mesh, 1)
    assert np.abs(small.p - (2.0 - xp[:, 0])).max() < 1e-9


def test_one_dimensional_flows_match_their_reference_outputs():
    """tests/1d_flow.output:10,30 and tests/1d_flow_damped.output:10,30,39,47,55 (NavierStokesMatrix<1>,
    navier_stokes_matrix.cc:1210): the first residual of every time step -- the only reference outputs that exercise the
    damping term (:831-835) and tau grad div; the undamped run drops to round-off after two steps (the reference prints
    its solver noise there), the damped one keeps printing 2e-6"""
    import pytest
    for case in ("1d_flow", "1d_flow_damped"):
        ref = _ref(case)
        fp = adaflo_amd.FlowParameters(velocity_degree=2, viscosity=ref["viscosity"], damping=ref["damping"],
                                       tau_grad_div=ref["tau_grad_div"], time_step_size_start=ref["dt"], end_time=1.0)
        sim = co.Flow1D(adaflo_amd.TimeStepping(fp), n=ref["cells"], viscosity=ref["viscosity"], damping=ref["damping"],
                        tau_grad_div=ref["tau_grad_div"])
        assert (sim.nu, sim.np_) == (ref["dofs_u"], ref["dofs_p"])
        for expected in ref["first_residuals"]:
            history = sim.advance_time_step()
            assert "%.3e" % history[0] == expected and history[-1] < 1e-11, (case, history)
        # incompressibility in 1D: the velocity stays uniform
        assert np.abs(sim.u - sim.u[0]).max() < 1e-10

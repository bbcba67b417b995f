"""GPU tier: integration of every row of the path -- a few time steps of the 3D rising bubble
(tests/rising_bubble_ls.prm extruded to 3D: [0,1]^2 x [0,2], bubble of radius 0.25) entirely on the
device: level-set advection / reinitialisation / normal / curvature solves, Heaviside + force with
the variable density / viscosity arrays, two-phase residual, Newton with FGMRES + block
preconditioner on the two-phase Jacobian.  The reference has no 3D golden output for this case
(its rising_bubble outputs are 2D), so the checks are physical invariants, plus a comparison of the
printed per-step quantities with the oracle's time step, which is pinned to the 2D golden output."""
import numpy as np
import pytest

import adaflo_amd
from adaflo_amd.level_set_okz_solver import LevelSetOKZSolver

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("exact_projection", [True, False])
def test_rising_bubble_three_time_steps(exact_projection):
    """exact_projection: normal / curvature projections by fast diagonalisation (the driver's default) or by
    diagonally preconditioned CG to the reference's tolerances; the centre of mass of both runs agrees to 1e-6"""
    fp = adaflo_amd.FlowParameters(
        velocity_degree=2, density=1.0, density_diff=-0.9, viscosity=0.01, viscosity_diff=-0.009,
        surface_tension=0.0245, gravity=0.98, epsilon=1.5, concentration_subdivisions=2,
        interpolate_grad_onto_pressure=True, curvature_correction=True, time_step_size_start=0.02, end_time=1.0,
        max_nl_iteration=10, tol_nl_iteration=1e-8, max_lin_iteration=200, tol_lin_iteration=1e-4)
    mesh = adaflo_amd.BrickMesh([8, 8, 16], [0., 0., 0.], [1., 1., 2.])
    centre = np.array([0.5, 0.5, 0.5])
    solver = LevelSetOKZSolver(fp, mesh, lambda x: np.linalg.norm(x - centre, axis=1) - 0.25,
                               exact_projection=exact_projection)
    solver.ops.compute_heaviside(solver.heaviside, solver.solution, fp.epsilon)
    vol0, c0 = solver.bubble_volume_and_centre()
    assert np.allclose(c0, centre, atol=1e-10)
    zs, vols = [c0[2]], [vol0]
    for step in range(3):
        n_newton, n_linear = solver.advance_time_step()
        ns = solver.navier_stokes
        assert np.hypot(*ns.history[-1]) < fp.tol_nl_iteration, (step, ns.history)
        assert n_newton <= fp.max_nl_iteration
        vol, c = solver.bubble_volume_and_centre()
        zs.append(c[2])
        vols.append(vol)
        rho, mu, _ = ns.navier_stokes_matrix.get_coefficients()
        assert 0.1 - 1e-12 <= rho.min() and rho.max() <= 1.0 + 1e-12
        assert 0.001 - 1e-12 <= mu.min() and mu.max() <= 0.01 + 1e-12
        phi = solver.solution.numpy()
        assert np.abs(phi).max() < 1.05
    # buoyancy: the velocity inside the bubble points upwards and the bubble starts to rise
    u = solver.navier_stokes.solution[0].cpu().numpy().reshape(-1, 3)
    assert u[:, 2].max() > 1e-3 and u[:, 2].max() > 5 * np.abs(u[:, :2]).max() * 0.1
    assert zs[-1] > zs[0] + 1e-6
    assert np.allclose(np.array(zs)[None, :] * 0 + c0[0], c0[0])            # (symmetry in x, y is checked below)
    _, c = solver.bubble_volume_and_centre()
    assert abs(c[0] - 0.5) < 1e-8 and abs(c[1] - 0.5) < 1e-8
    assert abs(vols[-1] - vols[0]) < 0.02 * vols[0]
    # both projection solvers give the same rise (the CG run stops at relative residuals of 1e-7 / 1e-8)
    seen = test_rising_bubble_three_time_steps.__dict__.setdefault("rise", {})
    seen[exact_projection] = zs[-1]
    if len(seen) == 2:
        assert abs(seen[True] - seen[False]) < 1e-6 * abs(seen[True])


@pytest.mark.parametrize("linearization,lin,s,max_nl,n_steps,n",
                         [("coupled implicit Newton", 0, 2, 10, 2, 8), ("coupled implicit Picard", 1, 3, 10, 2, 6),
                          ("coupled velocity semi-implicit", 2, 2, 1, 3, 6), ("coupled velocity explicit", 3, 3, 1, 2, 6)])
def test_device_time_steps_equal_the_oracle_time_steps_in_3d(linearization, lin, s, max_nl, n_steps, n):
    """The oracle's two-phase time step (oracle/two_phase_oracle.py) reproduces the reference's 2D
    golden outputs for all four treatments of the convective term (tests/test_oracle_golden_ls.py:
    rising_bubble_ls{,_picard,_imex,_expl}.output).  The same oracle algorithm in 3D is the checker
    here: on an 8 x 8 x 16 (6 x 6 x 12) mesh the device drivers (adaflo_amd.LevelSetOKZSolver) must print the same
    advection residual / iterations, reinitialisation iterations and first Navier-Stokes residual for
    the first two or three time steps (the CPU oracle dominates the run time).  (Start-of-step quantities only: the device solves its linear systems
    with FGMRES, the oracle exactly.)"""
    from threadpoolctl import threadpool_limits

    from oracle import two_phase_oracle as tpo
    kw = dict(velocity_degree=2, density=1.0, density_diff=-0.9, viscosity=0.01, viscosity_diff=-0.009,
              surface_tension=0.0245, gravity=0.98, epsilon=1.5, concentration_subdivisions=s,
              interpolate_grad_onto_pressure=True, curvature_correction=True, time_step_size_start=0.02, end_time=1.0,
              linearization=linearization, max_nl_iteration=max_nl, tol_nl_iteration=1e-9, max_lin_iteration=200,
              tol_lin_iteration=1e-4)
    fp = adaflo_amd.FlowParameters(**kw)
    mesh = adaflo_amd.BrickMesh([n, n, 2 * n], [0., 0., 0.], [1., 1., 2.])
    dev = LevelSetOKZSolver(fp, mesh, lambda x: np.linalg.norm(x - 0.5, axis=1) - 0.25)
    with threadpool_limits(limits=1, user_api="blas"):
        ref = tpo.RisingBubble(lambda: adaflo_amd.TimeStepping(adaflo_amd.FlowParameters(**kw)), ncell=(n, n, 2 * n), s=s,
                               no_slip_everywhere=True, linearization=lin, max_nl=max_nl)
        assert dev.initial_reinit_iterations == ref.log["initial_reinitialize"]
        for step in range(n_steps):
            (adv_r0, adv_it), rei_its, history = ref.advance_time_step()
            dev.navier_stokes.history.clear()
            dev.advance_time_step()
            d_it, d_r0 = dev.concentration_iterations[-1]
            assert abs(d_it - adv_it) <= 1 and abs(d_r0 - adv_r0) <= 1e-6 * max(adv_r0, 1e-10), (step, d_it, adv_it, d_r0, adv_r0)
            assert dev.reinit_iterations[-1] == rei_its, (step, dev.reinit_iterations[-1], rei_its)
            first = float(np.hypot(*dev.navier_stokes.history[0]))
            assert abs(first - history[0]) < 1e-6 * history[0], (step, first, history[0])
            # the converged flow of the step (the oracle's solutions are pinned to the reference's printed bubble
            # statistics in 2D): velocity, and pressure up to its constant
            u_dev = dev.navier_stokes.solution[0].cpu().numpy()
            p_dev = dev.navier_stokes.solution[1].cpu().numpy()
            assert rel(u_dev, ref.u) < 1e-5, (step, rel(u_dev, ref.u))
            assert rel(p_dev - p_dev.mean(), ref.p - ref.p.mean()) < 1e-5, (step,)
    assert rel(dev.solution.numpy(), ref.phi) < 1e-6


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def test_device_start_of_step_equals_the_oracle_for_q3_q2_in_3d():
    """Taylor-Hood Q3/Q2 in the two-phase driver (the oracle reproduces tests/rising_bubble_ls_q3.output in
    2D): everything the reference prints before the first linear solve of time step #1 -- initial and
    in-step reinitialisation iterations, advection, first residual = surface tension + gravity integrated
    against cubic test functions with the 4-point Gauss rule -- device against oracle on 6 x 6 x 12 cells."""
    from threadpoolctl import threadpool_limits

    from oracle import two_phase_oracle as tpo
    kw = dict(velocity_degree=3, density=1.0, density_diff=-0.9, viscosity=0.01, viscosity_diff=-0.009,
              surface_tension=0.0245, gravity=0.98, epsilon=1.5, concentration_subdivisions=2,
              interpolate_grad_onto_pressure=True, curvature_correction=True, time_step_size_start=0.02, end_time=1.0,
              max_nl_iteration=10, tol_nl_iteration=1e-9, max_lin_iteration=200, tol_lin_iteration=1e-4)
    fp = adaflo_amd.FlowParameters(**kw)
    mesh = adaflo_amd.BrickMesh([6, 6, 12], [0., 0., 0.], [1., 1., 2.])
    dev = LevelSetOKZSolver(fp, mesh, lambda x: np.linalg.norm(x - 0.5, axis=1) - 0.25)
    with threadpool_limits(limits=1, user_api="blas"):
        ref = tpo.RisingBubble(lambda: adaflo_amd.TimeStepping(adaflo_amd.FlowParameters(**kw)), ncell=(6, 6, 12), s=2, k=3,
                               no_slip_everywhere=True, max_nl=0)
        assert dev.initial_reinit_iterations == ref.log["initial_reinitialize"]
        (adv_r0, adv_it), rei_its, history = ref.advance_time_step()
    dev.init_time_advance()
    dev.advance_concentration()
    dev.reinitialize(dev.n_reinit_steps)
    dev.compute_force()
    first = dev.navier_stokes.compute_residual()
    assert dev.concentration_iterations[-1][0] == adv_it and dev.reinit_iterations[-1] == rei_its
    assert abs(first - history[0]) < 1e-6 * history[0], (first, history[0])
    assert rel(dev.solution.numpy(), ref.phi) < 1e-6
    assert rel(dev.curvature.numpy(), ref.kappa) < 1e-5


def test_robustness_paths_of_the_two_phase_time_step():
    """LevelSetBaseAlgorithm::advance_time_step (level_set_base.cc:262-278): a residual that doubles triggers ten
    extra diffusion steps and a second force / residual evaluation; get_concentration_range (two_phase_base.cc:515-545)
    feeds the three extra diffusion steps of reinitialize (reinitialization.cc:281-284); the sub-steppers follow the
    global step size (advance_concentration.cc:508, reinitialization.cc:264)"""
    fp = adaflo_amd.FlowParameters(
        velocity_degree=2, density=1.0, density_diff=-0.9, viscosity=0.01, viscosity_diff=-0.009,
        surface_tension=0.0245, gravity=0.98, epsilon=1.5, concentration_subdivisions=2,
        interpolate_grad_onto_pressure=True, time_step_size_start=0.02, time_step_size_min=0.0, end_time=1.0,
        max_nl_iteration=10, tol_nl_iteration=1e-8, max_lin_iteration=200, tol_lin_iteration=1e-4)
    mesh = adaflo_amd.BrickMesh([6, 6, 12], [0., 0., 0.], [1., 1., 2.])
    centre = np.array([0.5, 0.5, 0.5])
    solver = LevelSetOKZSolver(fp, mesh, lambda x: np.linalg.norm(x - centre, axis=1) - 0.25)
    # range of the level set on the iterated trapezoid points against a brute-force evaluation
    lo, hi = solver.get_concentration_range()
    s = 2
    nn = [s * n + 1 for n in mesh.ncell]
    phi = solver.solution.numpy().reshape(nn[2], nn[1], nn[0])
    t = np.arange(s + 3) / (s + 2.0)
    best = [np.inf, -np.inf]
    for cz in range(mesh.ncell[2]):
        for cy in range(mesh.ncell[1]):
            for cx in range(mesh.ncell[0]):
                loc = phi[s * cz:s * cz + s + 1, s * cy:s * cy + s + 1, s * cx:s * cx + s + 1]
                ax = np.arange(s + 1) / s
                v = loc
                for axis in range(3):
                    v = np.apply_along_axis(lambda line: np.interp(t, ax, line), axis, v)
                best = [min(best[0], v.min()), max(best[1], v.max())]
    assert abs(lo - best[0]) < 1e-14 and abs(hi - best[1]) < 1e-14 and -1.0 <= lo < -0.9 and 0.5 < hi <= 1.0
    # an out-of-range profile adds three diffusion steps to the next reinitialisation
    solver.advance_time_step()
    n_regular = len(solver.reinit_iterations[-1])
    solver.last_concentration_range = (-1.05, 1.0)
    solver.reinitialize(2)
    assert len(solver.reinit_iterations[-1]) == n_regular + 3
    solver.last_concentration_range = (lo, hi)
    # the sub-steppers follow a changed global step size
    solver.time_stepping.set_desired_time_step(0.015)
    solver.advance_time_step()
    assert abs(solver.ts_advect.step_size() - 0.015) < 1e-15 and abs(solver.ts_reinit.step_size() - 0.015) < 1e-15
    # excessive residual: pretend the previous step had a tiny residual
    for _ in range(3):
        solver.advance_time_step()
    assert solver.smoothing_steps == []
    solver.old_residual *= 1e-3
    n_before = len(solver.reinit_iterations)
    solver.advance_time_step()
    assert solver.smoothing_steps == [solver.time_stepping.step_no()]
    assert len(solver.reinit_iterations) == n_before + 2 and len(solver.reinit_iterations[-1]) >= 10
    ns = solver.navier_stokes
    assert np.hypot(*ns.history[-1]) < fp.tol_nl_iteration

#!/usr/bin/env python3
"""3D rising bubble (tests/rising_bubble_ls.prm of the reference, extruded: [0,1]^2 x [0,2], bubble of
radius 0.25 at (0.5, 0.5, 0.5)) with every kernel and every vector on one MI355X.

    python examples/rising_bubble_3d.py [cells_x] [subdivisions] [time steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adaflo_amd  # noqa: E402
from adaflo_amd.level_set_okz_solver import LevelSetOKZSolver  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    fp = adaflo_amd.FlowParameters(
        velocity_degree=2, density=1.0, density_diff=-0.9, viscosity=0.01, viscosity_diff=-0.009,
        surface_tension=0.0245, gravity=0.98, epsilon=1.5, concentration_subdivisions=s,
        interpolate_grad_onto_pressure=True, curvature_correction=True, time_step_size_start=0.02, end_time=3.0,
        max_nl_iteration=10, tol_nl_iteration=1e-9, max_lin_iteration=200, tol_lin_iteration=1e-4)
    mesh = adaflo_amd.BrickMesh([n, n, 2 * n], [0., 0., 0.], [1., 1., 2.])
    centre = np.array([0.5, 0.5, 0.5])
    solver = LevelSetOKZSolver(fp, mesh, lambda x: np.linalg.norm(x - centre, axis=1) - 0.25)
    m = solver.navier_stokes.navier_stokes_matrix
    print("cells %d, dofs velocity/pressure/level set: %d / %d / %d" % (mesh.n_cells, m.n_dofs_u(), m.n_dofs_p(), solver.ops.n_dofs))
    solver.ops.compute_heaviside(solver.heaviside, solver.solution, fp.epsilon)
    vol0, c0 = solver.bubble_volume_and_centre()
    for step in range(steps):
        import torch
        torch.cuda.synchronize()
        t0 = time.time()
        solver.navier_stokes.history.clear()
        solver.navier_stokes.linear_iterations.clear()
        n_newton, n_linear = solver.advance_time_step()
        torch.cuda.synchronize()
        vol, c = solver.bubble_volume_and_centre()
        ns = solver.navier_stokes
        umax = float(ns.solution[0].abs().max())
        print("step %2d t=%.3f  advect %s  reinit %s  newton %d (lin %d)  res %.2e -> %.2e  z_c %.6f  vol %.6f (%+.2e)  |u|max %.3e  %.2f s"
              % (step + 1, solver.time_stepping.now(), solver.concentration_iterations[-1][0], solver.reinit_iterations[-1],
                 n_newton, n_linear, np.hypot(*ns.history[0]), np.hypot(*ns.history[-1]), c[2], vol, vol / vol0 - 1, umax,
                 time.time() - t0), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""3D rising bubble (tests/rising_bubble.cc + rising_bubble_ls.prm of the reference, extruded:
[0,1]^2 x [0,2], bubble of radius 0.25 at (0.5, 0.5, 0.5), no-slip box) with every kernel and every
vector on one MI355X.  Prints the lines of the reference's output (verbosity 1):
    Concentration advance: advect [<initial residual>/<BiCGStab its>] and reinitialize (<CG its> + ...)
    Residual/iterations: [<residual>/<FGMRES its>] ... [<residual>/conv.]

    python examples/rising_bubble_3d.py [cells_x] [subdivisions] [time steps]
    python examples/rising_bubble_3d.py --prm case.prm [time steps]     # 5 * 2^"global refinements" cells in x

A parameter file with `set dimension = 2` (examples/rising_bubble_2d.prm = the values of the reference's
tests/rising_bubble_ls.prm) runs the reference's own 2D case -- [0,1] x [0,2], symmetry on the side walls, no-slip at the
bottom and the top (tests/rising_bubble.cc:119-150) -- on the device's flat-third-direction path; with 3 global
refinements (40 x 80 cells) the printed lines are those of tests/rising_bubble_ls.output."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adaflo_amd  # noqa: E402
from adaflo_amd.level_set_okz_solver import LevelSetOKZSolver  # noqa: E402


def main():
    args = sys.argv[1:]
    if args and args[0] == "--prm":
        with open(args[1]) as f:
            fp = adaflo_amd.flow_parameters_from_prm(f.read())
        n = 5 * 2 ** fp.global_refinements                    # rising_bubble.cc:125-133
        steps = int(args[2]) if len(args) > 2 else 5
    else:
        n = int(args[0]) if len(args) > 0 else 16
        s = int(args[1]) if len(args) > 1 else 4
        steps = int(args[2]) if len(args) > 2 else 5
        fp = adaflo_amd.FlowParameters(
            velocity_degree=2, density=1.0, density_diff=-0.9, viscosity=0.01, viscosity_diff=-0.009,
            surface_tension=0.0245, gravity=0.98, epsilon=1.5, concentration_subdivisions=s,
            interpolate_grad_onto_pressure=True, curvature_correction=True, time_step_size_start=0.02, end_time=3.0,
            max_nl_iteration=10, tol_nl_iteration=1e-9, max_lin_iteration=200, tol_lin_iteration=1e-4,
            n_reinit_steps=2, n_initial_reinit_steps=2)
    dim = 2 if fp.dimension == 2 else 3
    if dim == 2:
        mesh = adaflo_amd.BrickMesh([n, 2 * n], [0., 0.], [1., 2.])
        centre = np.array([0.5, 0.5, 0.0])
        symmetry = [0, 1]                                      # side walls (rising_bubble.cc:133-150)
    else:
        mesh = adaflo_amd.BrickMesh([n, n, 2 * n], [0., 0., 0.], [1., 1., 2.])
        centre = np.array([0.5, 0.5, 0.5])
        symmetry = []                                          # the extruded case is a no-slip box
    solver = LevelSetOKZSolver(fp, mesh, lambda x: np.linalg.norm(x - centre, axis=1) - 0.25,
                               n_reinit_steps=fp.n_reinit_steps, n_initial_reinit_steps=fp.n_initial_reinit_steps,
                               symmetry_faces=symmetry)
    m = solver.navier_stokes.navier_stokes_matrix
    print("Number of active cells: %d." % mesh.n_cells)
    nu_ = m.n_dofs_u() * dim // 3                              # (dim = 2: the engine carries a third, constrained component)
    print("Number of Navier-Stokes degrees of freedom: %d (%d + %d)." % (nu_ + m.n_dofs_p(), nu_, m.n_dofs_p()))
    print("Number of level set degrees of freedom: %d." % solver.ops.n_dofs)
    print("  reinitialize (%s)" % " + ".join(str(i) for i in solver.initial_reinit_iterations))
    if dim == 2:
        print("\n".join(solver.compute_bubble_statistics()["lines"]))
    solver.ops.compute_heaviside(solver.heaviside, solver.solution, fp.epsilon)
    vol0, c0 = solver.bubble_volume_and_centre()
    ts = solver.time_stepping
    for step in range(steps):
        import torch
        torch.cuda.synchronize()
        t0 = time.time()
        ns = solver.navier_stokes
        ns.history.clear()
        ns.linear_iterations.clear()
        solver.advance_time_step()
        torch.cuda.synchronize()
        wall = time.time() - t0
        vol, c = solver.bubble_volume_and_centre()
        print("\nTime step #%d, advancing from t_n-1 = %g to t = %g (dt = %g)." % (ts.step_no(), ts.previous(), ts.now(), ts.step_size()))
        it, r0 = solver.concentration_iterations[-1]
        print("  Concentration advance: advect [%.3g/%d] and reinitialize (%s)"
              % (r0, it, " + ".join(str(i) for i in solver.reinit_iterations[-1])))
        res = [float(np.hypot(*h)) for h in ns.history]
        its = [i for i, _ in ns.linear_iterations]
        print("  Residual/iterations: " + " ".join("[%.3g/%d]" % (r, i) for r, i in zip(res, its)) + " [%.3g/conv.]" % res[-1])
        if dim == 2:                                            # the reference's statistics (two_phase_base.cc:621-905)
            print("\n".join(solver.compute_bubble_statistics()["lines"]))
        else:
            print("  Position of the center of mass:  " + "  ".join("%.8g" % v for v in c[:dim]) + "   (lumped Heaviside)")
        print("  (bubble volume drift %+.2e, |u|max %.3e, %.2f s wall)" % (vol / vol0 - 1, float(ns.solution[0].abs().max()), wall), flush=True)


if __name__ == "__main__":
    main()

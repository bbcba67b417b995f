#!/usr/bin/env python3
"""wall-clock per phase of the two-phase time step (host gaps included), one MI355X"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adaflo_amd  # noqa: E402
from adaflo_amd.level_set_okz_solver import LevelSetOKZSolver  # noqa: E402
import torch  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    fp = adaflo_amd.FlowParameters(
        velocity_degree=2, density=1.0, density_diff=-0.9, viscosity=0.01, viscosity_diff=-0.009,
        surface_tension=0.0245, gravity=0.98, epsilon=1.5, concentration_subdivisions=s,
        interpolate_grad_onto_pressure=True, curvature_correction=True, time_step_size_start=0.02, end_time=3.0,
        max_nl_iteration=10, tol_nl_iteration=1e-9, max_lin_iteration=200, tol_lin_iteration=1e-4,
        iterations_before_inner_solvers=int(sys.argv[5]) if len(sys.argv) > 5 else 50)
    mesh = adaflo_amd.BrickMesh([n, n, 2 * n], [0., 0., 0.], [1., 1., 2.])
    solver = LevelSetOKZSolver(fp, mesh, lambda x: np.linalg.norm(x - 0.5, axis=1) - 0.25)
    ns = solver.navier_stokes
    if len(sys.argv) > 4:      # normal / curvature projections: 1 exact (fast diagonalisation, default), 0 CG
        solver.ops.exact_projection = bool(int(sys.argv[4]))
    if len(sys.argv) > 3:      # inner solves of the block preconditioner: 1 fast diagonalisation, 0 Jacobi
        ctx = ns.navier_stokes_matrix._require()
        adaflo_amd._lib.check(ctx, adaflo_amd._lib.load().adaflo_ns_preconditioner_set_inner(ctx, int(sys.argv[3])))

    if os.environ.get("ADAFLO_KERNEL_VARIANT"):   # e.g. 4: the Q2/Q1 vmult streams the state instead of recomputing it
        ns.navier_stokes_matrix.set_kernel_variant(int(os.environ["ADAFLO_KERNEL_VARIANT"]))
    if len(sys.argv) > 6:      # cheap stage of the two-phase solver: BiCGStab iterations of the velocity block (0 = off)
        ns.cheap_velocity_iterations = int(sys.argv[6])

    def timed(name, fn, acc):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r

    for step in range(3):
        acc = {}
        timed("init_time_advance", solver.init_time_advance, acc)
        timed("advance_concentration", solver.advance_concentration, acc)
        timed("reinitialize", lambda: solver.reinitialize(solver.n_reinit_steps), acc)
        timed("compute_force", solver.compute_force, acc)
        res = timed("compute_residual", ns.compute_residual, acc)
        p = fp
        for it in range(p.max_nl_iteration):
            tol = min(p.tol_lin_iteration * res, p.tol_lin_iteration)
            if res * p.tol_lin_iteration < 0.5 * p.tol_nl_iteration:
                tol = 0.5 * p.tol_nl_iteration
            if it == 0:
                timed("build_preconditioner", ns.build_preconditioner, acc)
            its, _ = timed("solve_system", lambda: ns.solve_system(tol), acc)
            acc["outer_its"] = acc.get("outer_its", 0) + its * 1e-3
            ns.solution[0] += ns.solution_update[0]
            ns.solution[1] += ns.solution_update[1]
            res = timed("compute_residual", ns.compute_residual, acc)
            if res < p.tol_nl_iteration:
                break
        import ctypes as C
        sv, iv = C.c_int64(), C.c_int64()
        adaflo_amd._lib.load().adaflo_ns_preconditioner_statistics(ns.navier_stokes_matrix._require(), C.byref(sv), C.byref(iv))
        print("step %d: total %.3f s | " % (step + 1, sum(acc.values())) + "  ".join("%s %.0f ms" % (k, 1e3 * v) for k, v in acc.items())
              + " | %d velocity solves, %.1f inner iterations each" % (sv.value, iv.value / max(sv.value, 1)), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Beltrami time steps on one MI355X: wall clock per step, outer FGMRES iterations and the inner velocity-block
iterations per preconditioner application, with the fast-diagonalisation inner solves (1) and with the
Jacobi-preconditioned inner Krylov solves (0).
usage: time_beltrami_step.py [cells_per_direction] [steps] [only this mode] [velocity degree]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adaflo_amd  # noqa: E402
from adaflo_amd import _lib, beltrami  # noqa: E402
from adaflo_amd.navier_stokes import NavierStokes, node_coordinates  # noqa: E402
import torch  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    nu = 1.0
    mesh = adaflo_amd.BrickMesh([n] * 3, [-1.0] * 3, [1.0] * 3)
    k = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    xu, xp = node_coordinates(mesh, k), node_coordinates(mesh, k - 1)
    modes = (int(sys.argv[3]),) if len(sys.argv) > 3 else (1, -1, 0)
    for inner in modes:             # 1: fast diagonalisation, two-stage solver; -1: inner solves from the start; 0: Jacobi
        fp = adaflo_amd.FlowParameters(velocity_degree=k, viscosity=nu, time_step_size_start=0.05 * 32 / (n * k), end_time=1.0,
                                       max_nl_iteration=10, tol_nl_iteration=1e-9, max_lin_iteration=100, tol_lin_iteration=1e-5,
                                       iterations_before_inner_solvers=0 if inner < 0 else 50)
        ns = NavierStokes(fp, mesh, adaflo_amd.TimeStepping(fp), dirichlet_function=lambda x, t: beltrami.velocity(x, t, nu))
        ctx = ns.navier_stokes_matrix._require()
        lib = _lib.load()
        _lib.check(ctx, lib.adaflo_ns_preconditioner_set_inner(ctx, abs(inner)))
        ns.set_initial_condition(beltrami.velocity(xu, 0.0, nu).reshape(-1), beltrami.pressure(xp, 0.0, nu))
        for step in range(steps):
            s0, i0 = C.c_int64(), C.c_int64()
            lib.adaflo_ns_preconditioner_statistics(ctx, C.byref(s0), C.byref(i0))
            ns.linear_iterations.clear()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ns.advance_time_step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            s1, i1 = C.c_int64(), C.c_int64()
            lib.adaflo_ns_preconditioner_statistics(ctx, C.byref(s1), C.byref(i1))
            ns_, ni = s1.value - s0.value, i1.value - i0.value
            print("inner=%d Q%d %d^3 step %d: %.3f s, outer iterations %s, %d velocity solves, %.1f inner iterations each"
                  % (inner, k, n, step + 1, dt, [i for i, _ in ns.linear_iterations], ns_, ni / max(ns_, 1)), flush=True)


if __name__ == "__main__":
    main()

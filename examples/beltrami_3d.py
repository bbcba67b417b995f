#!/usr/bin/env python3
"""The reference's 3D Beltrami test (tests/beltrami.cc, tests/beltrami_3d.prm: [-1,1]^3, nu = 1,
BDF-2, dt = 0.05, Q2/Q1, coupled implicit Newton) on one MI355X; prints the nonlinear residual
table of tests/beltrami_3d.output.      python examples/beltrami_3d.py [cells per direction] [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import adaflo_amd
from adaflo_amd.navier_stokes import NavierStokes, node_coordinates
from adaflo_amd import beltrami
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
fp = adaflo_amd.FlowParameters(velocity_degree=2, viscosity=1.0, time_step_size_start=0.05, end_time=1.0,
                               max_nl_iteration=10, tol_nl_iteration=1e-9, max_lin_iteration=100, tol_lin_iteration=1e-5)
mesh = adaflo_amd.BrickMesh([n] * 3, [-1.0] * 3, [1.0] * 3)
ns = NavierStokes(fp, mesh, adaflo_amd.TimeStepping(fp), dirichlet_function=lambda x, t: beltrami.velocity(x, t, 1.0))
xu, xp = node_coordinates(mesh, 2), node_coordinates(mesh, 1)
ns.set_initial_condition(beltrami.velocity(xu, 0.0, 1.0).reshape(-1), beltrami.pressure(xp, 0.0, 1.0))
import time, torch
for step in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    torch.cuda.synchronize(); t0 = time.time()
    print(ns.advance_time_step())
    torch.cuda.synchronize(); print("time step wall", time.time() - t0)
    for h in ns.history: print("   %-11.3e %-12.3e" % h)
    print(ns.linear_iterations)
    ns.history.clear(); ns.linear_iterations.clear()

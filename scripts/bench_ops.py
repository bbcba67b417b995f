#!/usr/bin/env python3
"""Secondary measurements (not the driver's bench contract): generic-degree NS vmult and the
level-set operators on one MI355X; prints one JSON line per operator with MDoF/s and the
algorithmic HBM rate (SURVEY.md 8d byte counts)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import adaflo_amd  # noqa: E402
from adaflo_amd import level_set_okz as lso  # noqa: E402


def timeit(fn, sync, reps=20, warm=3):
    for _ in range(warm):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps


def ns_case(k, n, variant, two_phase=False, state_from_residual=False, linearization=None):
    """state_from_residual: the state comes from NavierStokesMatrix::residual of the same context, the way a Newton
    step produces it -- for Q2/Q1 with Newton linearisation the vmult then recomputes it from the nodal linearisation
    point (kernel variant 1; variant 4 streams it); otherwise a random canonical array is set and streamed"""
    fp = adaflo_amd.FlowParameters(velocity_degree=k, density_diff=0.5 if two_phase else 0.0)
    if linearization is not None:                    # (index into parameters.LINEARIZATIONS)
        from adaflo_amd.parameters import LINEARIZATIONS
        fp.linearization = {v: n for n, v in LINEARIZATIONS.items()}[linearization]
    ts = adaflo_amd.TimeStepping(fp)
    for _ in range(3):
        ts.next()
    op = adaflo_amd.NavierStokesMatrix(fp, adaflo_amd.BrickMesh([n] * 3, [-1] * 3, [1] * 3))
    op.initialize(ts, True)
    op.set_kernel_variant(variant)
    rng = np.random.default_rng(1)
    nq = (k + 1) ** 3
    if not state_from_residual:
        op.set_linearization(rng.uniform(-1, 1, op.n_cells() * nq * 12))
    if two_phase:  # variable rho, mu, damping at the quadrature points
        nc = op.n_cells() * nq
        op.set_coefficients(rng.uniform(.5, 2, nc), rng.uniform(.5, 2, nc), rng.uniform(-.5, .5, nc))
    src = op.block_vector(rng.uniform(-1, 1, op.n_dofs_u()), rng.uniform(-1, 1, op.n_dofs_p()))
    dst = op.block_vector()
    if state_from_residual:
        op.residual(dst, src, None, op.block_vector(rng.uniform(-1, 1, op.n_dofs_u())),
                    op.block_vector(rng.uniform(-1, 1, op.n_dofs_u())))
    t = timeit(lambda: op.vmult(dst, src), op.synchronize)
    ndof = op.n_dofs_u() + op.n_dofs_p()
    b_alg = op.n_cells() * (16 * (3 * k ** 3 + (k - 1) ** 3) + 8 * (15 if two_phase else 12) * nq)
    print(json.dumps({"op": "ns_vmult" + ("_two_phase" if two_phase else "") + ("_after_residual" if state_from_residual else "")
                      + ("" if linearization is None else "_lin%d" % linearization),
                      "k": k, "cells": n, "variant": variant, "ms": round(t * 1e3, 4),
                      "MDoF/s": round(ndof / t / 1e6, 1), "alg_GB/s": round(b_alg / t / 1e9, 1),
                      "frac_of_8TB/s": round(b_alg / t / 8e12, 4)}), flush=True)


def ns_residual_case(k, n, variant, two_phase=False, lazy=True, linearization=None):
    """NavierStokesMatrix::residual (a3: the producer of the q-point state, once per Newton step);
    algorithmic bytes: 3 velocity vectors + p read, 2 vectors written, state written once (two_phase: + rho, mu, damping
    read per quadrature point)"""
    fp = adaflo_amd.FlowParameters(velocity_degree=k, density_diff=0.5 if two_phase else 0.0)
    if linearization is not None:                    # (index into parameters.LINEARIZATIONS: 4 = projection)
        from adaflo_amd.parameters import LINEARIZATIONS
        fp.linearization = {v: n for n, v in LINEARIZATIONS.items()}[linearization]
    ts = adaflo_amd.TimeStepping(fp)
    for _ in range(3):
        ts.next()
    op = adaflo_amd.NavierStokesMatrix(fp, adaflo_amd.BrickMesh([n] * 3, [-1] * 3, [1] * 3))
    op.initialize(ts, True)
    op.set_kernel_variant(variant)
    if k == 2 and variant:
        op.set_lazy_state(lazy)      # (lazy: the state is laid out only when somebody asks; the bytes below still count it)
    rng = np.random.default_rng(1)
    if two_phase:
        nc = op.n_cells() * (k + 1) ** 3
        op.set_coefficients(rng.uniform(.5, 2, nc), rng.uniform(.5, 2, nc), rng.uniform(-.5, .5, nc))
    sol = op.block_vector(rng.uniform(-1, 1, op.n_dofs_u()), rng.uniform(-1, 1, op.n_dofs_p()))
    old = adaflo_amd.BlockVector([op.initialize_u_vector(rng.uniform(-1, 1, op.n_dofs_u()))])
    oldold = adaflo_amd.BlockVector([op.initialize_u_vector(rng.uniform(-1, 1, op.n_dofs_u()))])
    rhs = op.block_vector()
    t = timeit(lambda: op.residual(rhs, sol, None, old, oldold), op.synchronize, reps=10, warm=2)
    nq = (k + 1) ** 3
    b_alg = op.n_cells() * (8 * (5 * 3 * k ** 3 + 2 * (k - 1) ** 3) + 8 * (15 if two_phase else 12) * nq)
    print(json.dumps({"op": "ns_residual" + ("_two_phase" if two_phase else "") + ("" if linearization is None else "_lin%d" % linearization),
                      "k": k, "cells": n, "variant": variant,
                      "state": "lazy" if (lazy and k == 2 and variant and linearization in (None, 0)) else "written", "ms": round(t * 1e3, 4),
                      "alg_GB/s": round(b_alg / t / 1e9, 1), "frac_of_8TB/s": round(b_alg / t / 8e12, 4)}), flush=True)


def ns_divergence_case(ncell, variant):
    """divergence_vmult_add (a5), once per application of the block preconditioner: reads the
    velocity vector, read-modify-writes the pressure vector"""
    fp = adaflo_amd.FlowParameters(velocity_degree=2)
    ts = adaflo_amd.TimeStepping(fp)
    for _ in range(3):
        ts.next()
    op = adaflo_amd.NavierStokesMatrix(fp, adaflo_amd.BrickMesh(list(ncell), [0, 0, 0], [1, 1, 2]))
    op.initialize(ts, True)
    op.set_kernel_variant(variant)
    rng = np.random.default_rng(1)
    su = op.initialize_u_vector(rng.uniform(-1, 1, op.n_dofs_u()))
    dp = op.initialize_p_vector()
    t = timeit(lambda: op.divergence_vmult_add(dp, su, False), op.synchronize)
    b_alg = 8 * op.n_dofs_u() + 16 * op.n_dofs_p()
    print(json.dumps({"op": "ns_divergence_vmult_add", "cells": list(ncell), "variant": variant, "ms": round(t * 1e3, 4),
                      "alg_GB/s": round(b_alg / t / 1e9, 1), "frac_of_8TB/s": round(b_alg / t / 8e12, 4)}), flush=True)


def ns_host_vector_case(n):
    """adapter case of SURVEY 8d: src / dst live in (pinned) host memory, so every vmult pays
    H2D of src and D2H of dst over PCIe -- reported beside, never instead of, the resident rate"""
    import torch
    fp = adaflo_amd.FlowParameters(velocity_degree=2)
    ts = adaflo_amd.TimeStepping(fp)
    for _ in range(3):
        ts.next()
    stream = torch.cuda.current_stream().cuda_stream
    op = adaflo_amd.NavierStokesMatrix(fp, adaflo_amd.BrickMesh([n] * 3, [-1] * 3, [1] * 3), stream=stream)
    op.initialize(ts, True)
    rng = np.random.default_rng(1)
    op.set_linearization(rng.uniform(-1, 1, op.n_cells() * 27 * 12))
    nu, npp = op.n_dofs_u(), op.n_dofs_p()
    h_src = [torch.from_numpy(rng.uniform(-1, 1, m)).pin_memory() for m in (nu, npp)]
    h_dst = [torch.empty(m, dtype=torch.float64).pin_memory() for m in (nu, npp)]
    d_src = [torch.empty(m, dtype=torch.float64, device="cuda") for m in (nu, npp)]
    d_dst = [torch.empty(m, dtype=torch.float64, device="cuda") for m in (nu, npp)]
    src = adaflo_amd.BlockVector([op.wrap(t) for t in d_src])
    dst = adaflo_amd.BlockVector([op.wrap(t) for t in d_dst])

    def step():
        for d, h in zip(d_src, h_src):
            d.copy_(h, non_blocking=True)
        op.vmult(dst, src)
        for d, h in zip(d_dst, h_dst):
            h.copy_(d, non_blocking=True)
    t = timeit(step, torch.cuda.synchronize, reps=10)
    print(json.dumps({"op": "ns_vmult_host_vectors (pinned, PCIe-inclusive)", "k": 2, "cells": n,
                      "ms": round(t * 1e3, 3), "MDoF/s": round((nu + npp) / t / 1e6, 1),
                      "pcie_GB/s": round(16 * (nu + npp) / t / 1e9, 1)}), flush=True)


def ls_case(s, ncell, only=None):
    mesh = adaflo_amd.BrickMesh(list(ncell), [0, 0, 0], [1, 1, 2])
    ops = lso.LevelSetOperators(mesh, s)
    ops.set_parameters(1.5 * max(mesh.h) / s, 0.02, 75.0, -100.0, 25.0, 1.5)
    rng = np.random.default_rng(2)
    nq = (2 * s) ** 3
    ncells = int(np.prod(ncell))
    src = ops.vector(rng.uniform(-1, 1, ops.n_dofs))
    dst = ops.vector()
    sync = lambda: adaflo_amd._lib.load().adaflo_synchronize(ops._ctx)
    adv = lso.LevelSetOKZSolverAdvanceConcentration(ops)
    adv.evaluated_convection = rng.uniform(-1, 1, ncells * nq * 3)
    rei = lso.LevelSetOKZSolverReinitialization(ops)
    rei.evaluated_normal = rng.uniform(-1, 1, ncells * nq * 3)
    nor = lso.LevelSetOKZSolverComputeNormal(ops)
    cur = lso.LevelSetOKZSolverComputeCurvature(ops)
    src3, dst3 = ops.vector(rng.uniform(-1, 1, 3 * ops.n_dofs), blocks=3), ops.vector(blocks=3)
    cases = [("ls_advect_vmult", lambda: adv.advance_concentration_vmult(dst, src), 1, 16 * s ** 3 + 24 * nq),
             ("ls_reinit_vmult", lambda: rei.reinitialization_vmult(dst, src, False), 1, 16 * s ** 3 + 24 * nq),
             ("ls_normal_vmult", lambda: nor.compute_normal_vmult(dst3, src3), 3, 3 * 16 * s ** 3),
             ("ls_curvature_vmult", lambda: cur.compute_curvature_vmult(dst, src, True), 1, 16 * s ** 3)]
    # right-hand sides (velocity of degree 2 needs a context with the flow spaces)
    ops_v = lso.LevelSetOperators(mesh, s, velocity_degree=2)
    ops_v.set_parameters(1.5 * max(mesh.h) / s, 0.02, 75.0, -100.0, 25.0, 1.5)
    adv_v, rei_v = lso.LevelSetOKZSolverAdvanceConcentration(ops_v), lso.LevelSetOKZSolverReinitialization(ops_v)
    nu = int(np.prod([2 * n + 1 for n in ncell]))
    vel = adaflo_amd.DeviceVector.from_numpy(ops_v._ctx, rng.uniform(-1, 1, 3 * nu))
    phi, old, oo, rhs = (ops_v.vector(rng.uniform(-1, 1, ops_v.n_dofs)) for _ in range(4))
    nrm = ops_v.vector(rng.uniform(-1, 1, 3 * ops_v.n_dofs), blocks=3)
    sync_v = lambda: adaflo_amd._lib.load().adaflo_synchronize(ops_v._ctx)
    nodal = 8 * s ** 3          # bytes per cell of one nodal field
    # (the advection right-hand side of the sweep structure no longer writes evaluated_convection: 24 nq less)
    rhs_cases = [("ls_advect_rhs", lambda: adv_v.local_advance_concentration_rhs(rhs, phi, old, oo, vel, True), 5 * nodal + 3 * 8 * 8),
                 ("ls_reinit_rhs_first", lambda: rei_v.local_reinitialize_rhs(rhs, phi, nrm, False, True), 6 * nodal + 24 * nq),
                 ("ls_reinit_rhs", lambda: rei_v.local_reinitialize_rhs(rhs, phi, nrm, False, False), 3 * nodal + 24 * nq),
                 ("ls_reinit_rhs_diffuse", lambda: rei_v.local_reinitialize_rhs(rhs, phi, nrm, True, False), 3 * nodal)]
    if only is not None:
        cases = [c for c in cases if c[0] in only]
        rhs_cases = [c for c in rhs_cases if c[0] in only]
    # the advection operator right after a sweep right-hand side: the velocity is evaluated from the nodal field the
    # engine kept (Q1_ADVECT_NODAL); algorithmic bytes: two nodal fields + the velocity nodes of a cell
    if only is None or "ls_advect_vmult_nodal" in only:
        ops_v.set_kernel_variant(1)
        adv_v.local_advance_concentration_rhs(rhs, phi, old, oo, vel, True)
        t = timeit(lambda: adv_v.advance_concentration_vmult(rhs, phi), sync_v)
        b = 2 * nodal + 3 * 8 * 8
        print(json.dumps({"op": "ls_advect_vmult_nodal", "s": s, "cells": list(ncell), "ms": round(t * 1e3, 4),
                          "MDoF/s": round(ops_v.n_dofs / t / 1e6, 1), "alg_GB/s": round(b * ncells / t / 1e9, 1),
                          "frac_of_8TB/s": round(b * ncells / t / 8e12, 4)}), flush=True)
    # the reinitialisation operator right after a first-step right-hand side: unit normal recomputed from the nodal field
    if only is None or "ls_reinit_vmult_nodal" in only:
        ops_v.set_kernel_variant(1)
        rei_v.local_reinitialize_rhs(rhs, phi, nrm, False, True)
        t = timeit(lambda: rei_v.reinitialization_vmult(rhs, phi, False), sync_v)
        b = 5 * nodal
        print(json.dumps({"op": "ls_reinit_vmult_nodal", "s": s, "cells": list(ncell), "ms": round(t * 1e3, 4),
                          "MDoF/s": round(ops_v.n_dofs / t / 1e6, 1), "alg_GB/s": round(b * ncells / t / 1e9, 1),
                          "frac_of_8TB/s": round(b * ncells / t / 8e12, 4)}), flush=True)
    for variant in (1, 0):
        ops_v.set_kernel_variant(variant)
        for name, fn, bytes_per_cell in rhs_cases:
            t = timeit(fn, sync_v)
            print(json.dumps({"op": name, "variant": variant, "s": s, "cells": list(ncell), "ms": round(t * 1e3, 4),
                              "alg_GB/s": round(bytes_per_cell * ncells / t / 1e9, 1),
                              "frac_of_8TB/s": round(bytes_per_cell * ncells / t / 8e12, 4)}), flush=True)
    for name, fn, blocks, bytes_per_cell in cases:
        t = timeit(fn, sync)
        print(json.dumps({"op": name, "s": s, "cells": list(ncell), "ms": round(t * 1e3, 4),
                          "MDoF/s": round(blocks * ops.n_dofs / t / 1e6, 1),
                          "alg_GB/s": round(bytes_per_cell * ncells / t / 1e9, 1),
                          "frac_of_8TB/s": round(bytes_per_cell * ncells / t / 8e12, 4)}), flush=True)


def krylov_case(s, ncell):
    """device-resident CG on the curvature projection system (compute_curvature.cc:345-355) with a
    mass-type diagonal preconditioner: iterations and time per iteration"""
    from adaflo_amd import solvers
    mesh = adaflo_amd.BrickMesh(list(ncell), [0, 0, 0], [1, 1, 2])
    ops = lso.LevelSetOperators(mesh, s)
    ops.set_parameters(1.5 * max(mesh.h) / s, 0.02, 75.0, -100.0, 25.0, 1.5)
    rng = np.random.default_rng(3)
    b = ops.vector(rng.uniform(-1, 1, ops.n_dofs))
    # diagonal of the operator ~ row sums of the mass part: A * 1 (no constraints)
    ones, diag = ops.vector(np.ones(ops.n_dofs)), ops.vector()
    cur = lso.LevelSetOKZSolverComputeCurvature(ops)
    cur.compute_curvature_vmult(diag, ones, False)
    pre = solvers.DiagonalPreconditioner(diag)
    sync = lambda: adaflo_amd._lib.load().adaflo_synchronize(ops._ctx)
    for _ in range(2):
        control = solvers.ReductionControl(2000, 1e-50, 1e-8)
        x = ops.vector()
        sync()
        t0 = time.perf_counter()
        solvers.SolverCG(control).solve(solvers.ComputeCurvatureMatrix(ops), x, b, pre)
        sync()
        t = time.perf_counter() - t0
    print(json.dumps({"op": "ls_curvature_cg_solve", "s": s, "cells": list(ncell), "dofs": ops.n_dofs,
                      "iterations": control.last_step(), "ms": round(t * 1e3, 3),
                      "ms_per_iteration": round(t * 1e3 / max(control.last_step(), 1), 4),
                      "reduction": control.last_value() / control.initial_value()}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "residual":
        ns_residual_case(2, 128, 1)
        ns_residual_case(2, 128, 0)
        for v in (1, 0):
            ns_divergence_case((64, 64, 128), v)
            ns_divergence_case((128, 128, 128), v)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "div":
        for v in (1, 2, 0):
            ns_divergence_case((64, 64, 128), v)
            ns_divergence_case((128, 128, 128), v)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "stencil":
        ls_case(4, (40, 40, 80), only=("ls_normal_vmult", "ls_curvature_vmult"))
        ls_case(4, (64, 64, 128), only=("ls_normal_vmult", "ls_curvature_vmult"))
        krylov_case(4, (40, 40, 80))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ls":
        ls_case(4, (64, 64, 128))
        sys.exit(0)
    ns_case(2, 128, 1)
    ns_case(2, 128, 0)
    ns_residual_case(2, 128, 1)
    ns_residual_case(2, 128, 0)
    ns_residual_case(2, 128, 1, two_phase=True)      # round 5: variable coefficients on the sweep kernel
    ns_residual_case(2, 128, 0, two_phase=True)
    for v in (1, 2, 0):
        ns_divergence_case((64, 64, 128), v)
        ns_divergence_case((128, 128, 128), v)
    ns_host_vector_case(128)
    ns_case(2, 128, 1, two_phase=True)
    ns_case(2, 128, 0, two_phase=True)
    for v in (1, 4):                                 # round 5: state recomputed from the nodal linearisation point (1) / streamed (4)
        ns_case(2, 128, v, state_from_residual=True)
        ns_case(2, 128, v, two_phase=True, state_from_residual=True)
    has_v2 = bool(adaflo_amd._lib.load().adaflo_has_kernel_variant(2))   # (the superseded kernels: ADAFLO_BUILD_VARIANTS=1 only)
    for k, n in ((3, 64), (4, 64), (5, 48)):        # 1: x-marching kernel (round 4), 2: z-sweep kernel (round 2), 0: generic
        for v in ((1, 2, 0) if has_v2 else (1, 0)):
            ns_case(k, n, v)
    ns_residual_case(3, 64, 1)
    ns_residual_case(4, 64, 1)
    ns_residual_case(5, 48, 1)
    for k, n in ((3, 64), (4, 64)):                  # round 6: two-phase residual on the x-marching kernel / generic
        for v in (1, 0):
            ns_residual_case(k, n, v, two_phase=True)
    for k, n in ((2, 128), (4, 64)):                 # round 6: the projection scheme's residual on the sweep kernels / generic
        for v in (1, 0):
            ns_residual_case(k, n, v, linearization=4)
    for k in (3, 4):                                 # two-phase Jacobian on the x-marching kernel (round 4) / generic
        for v in (1, 0):
            ns_case(k, 64, v, two_phase=True)
    ls_case(4, (40, 40, 80))
    krylov_case(4, (40, 40, 80))

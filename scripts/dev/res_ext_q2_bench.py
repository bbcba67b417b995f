"""development: Q2/Q1 residual of the semi-implicit scheme on the sweep kernel (variant 1, template EXT) against the generic
kernel (variant 0), alone and followed by the first vmult (which re-lays out the state after the generic kernel)"""
import json, sys
sys.path.insert(0, "."); sys.path.insert(0, "scripts")
import numpy as np
import adaflo_amd
from bench_ops import timeit

for lin in ("coupled velocity semi-implicit", "coupled velocity explicit"):
    for variant in (1, 0):
        fp = adaflo_amd.FlowParameters(velocity_degree=2, linearization=lin)
        ts = adaflo_amd.TimeStepping(fp)
        for _ in range(3):
            ts.next()
        op = adaflo_amd.NavierStokesMatrix(fp, adaflo_amd.BrickMesh([128] * 3, [-1] * 3, [1] * 3))
        op.initialize(ts, True)
        op.set_kernel_variant(variant)
        rng = np.random.default_rng(1)
        sol = op.block_vector(rng.uniform(-1, 1, op.n_dofs_u()), rng.uniform(-1, 1, op.n_dofs_p()))
        old = adaflo_amd.BlockVector([op.initialize_u_vector(rng.uniform(-1, 1, op.n_dofs_u()))])
        oldold = adaflo_amd.BlockVector([op.initialize_u_vector(rng.uniform(-1, 1, op.n_dofs_u()))])
        rhs, dst = op.block_vector(), op.block_vector()
        t = timeit(lambda: op.residual(rhs, sol, None, old, oldold), op.synchronize, reps=10, warm=2)
        def both():
            op.set_kernel_variant(variant)
            op.residual(rhs, sol, None, old, oldold)
            op.set_kernel_variant(1)
            op.vmult(dst, sol)
        t2 = timeit(both, op.synchronize, reps=10, warm=2)
        print(json.dumps({"op": "ns_residual " + lin, "k": 2, "cells": 128, "variant": variant, "ms": round(t * 1e3, 4),
                          "ms_residual_plus_first_vmult_on_the_sweep_kernel": round(t2 * 1e3, 4)}), flush=True)

#!/usr/bin/env python3
"""time of one fast-diagonalisation application on the level-set grid and on the Q1 pressure grid of the two-phase
benchmark, with the fast cosine transforms and (ADAFLO_FDM_NO_DCT) with the matrix products"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import adaflo_amd
from adaflo_amd import _lib
from adaflo_amd import level_set_okz as lso

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
s = int(sys.argv[2]) if len(sys.argv) > 2 else 4
mesh = adaflo_amd.BrickMesh([n, n, 2 * n], [0.0] * 3, [1.0, 1.0, 2.0])
ops = lso.LevelSetOperators(mesh, s)
ops.set_parameters(1.5 * max(mesh.h) / s, 0.02, 75.0, -100.0, 25.0, 1.5)
lib, ctx = _lib.load(), ops._ctx
x = ops.vector(np.random.default_rng(1).uniform(-1, 1, ops.n_dofs))
y = ops.vector()




def timed(field, src, dst, cm, cl, reps=20):
    for _ in range(3):
        _lib.check(ctx, lib.adaflo_fdm_apply(ctx, field, dst, src, cm, cl))
    lib.adaflo_synchronize(ctx)
    t0 = time.perf_counter()
    for _ in range(reps):
        _lib.check(ctx, lib.adaflo_fdm_apply(ctx, field, dst, src, cm, cl))
    lib.adaflo_synchronize(ctx)
    return (time.perf_counter() - t0) / reps * 1e3


for label, env in (("cosine transforms", None), ("matrix products", "1")):
    if env:
        os.environ["ADAFLO_FDM_NO_DCT"] = env
    else:
        os.environ.pop("ADAFLO_FDM_NO_DCT", None)
    ms = timed(2, x.ptr, y.ptr, 1.0, 0.01)
    gb = 10 * ops.n_dofs * 8 / 1e9                      # five passes, read + write
    print("level-set grid %d x %d x %d, %-18s %.3f ms per application (%.0f GB/s of pass traffic)"
          % (s * n + 1, s * n + 1, 2 * s * n + 1, label + ":", ms, gb / ms * 1e3), flush=True)

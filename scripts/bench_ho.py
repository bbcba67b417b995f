#!/usr/bin/env python3
"""Timing of the Q3..Q5 kernels (development aid): variant 1 = x-marching kernel (ns_hox.hip), 2 = z-sweep kernel of
round 2 (ns_ho.hip), 0 = generic.   usage: bench_ho.py [k,k,...] [variants] [x-chunks] [cells per direction]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import adaflo_amd


def run(k, n, variant, phys="incompressible", lx=0):
    fp = adaflo_amd.FlowParameters(velocity_degree=k, physical_type=phys)
    ts = adaflo_amd.TimeStepping(fp)
    for _ in range(3):
        ts.next()
    op = adaflo_amd.NavierStokesMatrix(fp, adaflo_amd.BrickMesh([n] * 3, [-1] * 3, [1] * 3))
    op.initialize(ts, True)
    op.set_kernel_variant(variant)
    op.set_x_chunk(lx)
    rng = np.random.default_rng(1)
    nq = (k + 1) ** 3
    if phys != "stokes":
        op.set_linearization(rng.uniform(-1, 1, op.n_cells() * nq * 12))
    src = op.block_vector(rng.uniform(-1, 1, op.n_dofs_u()), rng.uniform(-1, 1, op.n_dofs_p()))
    dst = op.block_vector()
    for _ in range(3):
        op.vmult(dst, src)
    op.synchronize()
    op.get_kernel_statistics()
    t0 = time.perf_counter()
    for _ in range(20):
        op.vmult(dst, src)
    op.synchronize()
    t = (time.perf_counter() - t0) / 20
    ks, kc = op.get_kernel_statistics()
    print(json.dumps({"k": k, "cells": n, "variant": variant, "lx": lx, "phys": phys, "ms": round(t * 1e3, 4),
                      "kernel_ms": round(1e3 * ks / max(kc, 1), 4), "GDoF/s": round((op.n_dofs_u() + op.n_dofs_p()) / t / 1e9, 2)}), flush=True)


if __name__ == "__main__":
    ks = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4]
    variants = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 0]
    chunks = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
    for k in ks:
        n = int(sys.argv[4]) if len(sys.argv) > 4 else {3: 64, 4: 64, 5: 48}[k]
        for v in variants:
            for lx in (chunks if v == 1 else [0]):
                run(k, n, v, "incompressible stationary" if k == 4 else "incompressible", lx)
                if v:
                    run(k, n, v, "stokes", lx)

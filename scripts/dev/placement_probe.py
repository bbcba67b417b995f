#!/usr/bin/env python3
"""development probe: does memory allocated (and kept / freed) before the engine exists change the placement the Q2/Q1
kernel draws?  usage: placement_probe.py <GB kept> <GB allocated and freed>; prints the kernel time at 128^3"""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import adaflo_amd  # noqa: E402

keep_gb, free_gb = float(sys.argv[1]), float(sys.argv[2])
torch.cuda.init()
kept = torch.empty(int(keep_gb * 2**30), dtype=torch.uint8, device="cuda") if keep_gb > 0 else None
if free_gb > 0:
    tmp = torch.empty(int(free_gb * 2**30), dtype=torch.uint8, device="cuda")
    tmp.fill_(1)
    del tmp
    torch.cuda.empty_cache()
n = 128
rng = np.random.default_rng(1)
fp = adaflo_amd.FlowParameters(velocity_degree=2)
ts = adaflo_amd.TimeStepping(fp)
for _ in range(3):
    ts.next()
op = adaflo_amd.NavierStokesMatrix(fp, adaflo_amd.BrickMesh([n] * 3, [-1] * 3, [1] * 3))
op.initialize(ts, True)
op.set_linearization(rng.uniform(-1, 1, op.n_cells() * 27 * 12))
src = op.block_vector(rng.uniform(-1, 1, op.n_dofs_u()), rng.uniform(-1, 1, op.n_dofs_p()))
dst = op.block_vector()
for _ in range(10):
    op.vmult(dst, src)
op.synchronize()
op.get_kernel_statistics()
for _ in range(40):
    op.vmult(dst, src)
op.synchronize()
ksec, kcount = op.get_kernel_statistics()
print("kept %.0f GB, freed %.0f GB: kernel %.4f ms" % (keep_gb, free_gb, 1e3 * ksec / kcount), flush=True)

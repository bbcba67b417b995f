#!/usr/bin/env python3
"""development probe: kernel time of the Q2/Q1 vmult at 128^3 for several freshly created engine instances in
ONE process (is the run-to-run spread of bench.py a property of the process or of the allocation?)."""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import adaflo_amd  # noqa: E402

n = 128
rng = np.random.default_rng(1)
lin = None
keep = []
for inst in range(6):
    fp = adaflo_amd.FlowParameters(velocity_degree=2)
    ts = adaflo_amd.TimeStepping(fp)
    for _ in range(3):
        ts.next()
    stream = torch.cuda.current_stream().cuda_stream
    op = adaflo_amd.NavierStokesMatrix(fp, adaflo_amd.BrickMesh([n] * 3, [-1] * 3, [1] * 3), stream=stream)
    op.initialize(ts, True)
    if lin is None:
        lin = rng.uniform(-1, 1, op.n_cells() * 27 * 12)
    op.set_linearization(lin)
    nu, npp = op.n_dofs_u(), op.n_dofs_p()
    src = op.block_vector(rng.uniform(-1, 1, nu), rng.uniform(-1, 1, npp))
    dst = op.block_vector()
    for rep in range(2):
        if rep == 0:
            os.environ["ADAFLO_DEBUG_PTR_ONCE"] = "1"
        for _ in range(10):
            op.vmult(dst, src)
        op.synchronize()
        op.get_kernel_statistics()
        for _ in range(20):
            op.vmult(dst, src)
        op.synchronize()
        ksec, kcount = op.get_kernel_statistics()
        print("instance %d rep %d: kernel %.4f ms  (state ptr mod 2^30 = %#x)" % (inst, rep, 1e3 * ksec / kcount, 0), flush=True)
    if inst % 2 == 0:
        keep.append(torch.empty((inst + 1) * 37_000_001, dtype=torch.float64, device="cuda"))  # shift later allocations
    del op, src, dst

#!/usr/bin/env python3
"""development probe: one engine, one state; kernel time of the Q2/Q1 vmult at 128^3 for several separately
allocated dst / src vectors (which buffer's physical placement moves the kernel between 1.26 and 1.38 ms?)."""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import adaflo_amd  # noqa: E402

n = 128
rng = np.random.default_rng(1)
fp = adaflo_amd.FlowParameters(velocity_degree=2)
ts = adaflo_amd.TimeStepping(fp)
for _ in range(3):
    ts.next()
stream = torch.cuda.current_stream().cuda_stream
op = adaflo_amd.NavierStokesMatrix(fp, adaflo_amd.BrickMesh([n] * 3, [-1] * 3, [1] * 3), stream=stream)
op.initialize(ts, True)
op.set_linearization(rng.uniform(-1, 1, op.n_cells() * 27 * 12))
nu, npp = op.n_dofs_u(), op.n_dofs_p()
su0 = torch.from_numpy(rng.uniform(-1, 1, nu)).cuda()
sp = torch.from_numpy(rng.uniform(-1, 1, npp)).cuda()
dp = torch.empty(npp, dtype=torch.float64, device="cuda")
pad = []
dsts, srcs = [], []
for i in range(6):
    dsts.append(torch.empty(nu, dtype=torch.float64, device="cuda"))
    pad.append(torch.empty(13_000_003 * (i + 1), dtype=torch.float64, device="cuda"))
    srcs.append(su0.clone())


def run(s, d):
    src = adaflo_amd.BlockVector([op.wrap(s), op.wrap(sp)])
    dst = adaflo_amd.BlockVector([op.wrap(d), op.wrap(dp)])
    for _ in range(5):
        op.vmult(dst, src)
    op.synchronize()
    op.get_kernel_statistics()
    for _ in range(20):
        op.vmult(dst, src)
    op.synchronize()
    ksec, kcount = op.get_kernel_statistics()
    return 1e3 * ksec / kcount


for i, d in enumerate(dsts):
    print("dst %d (%#x), src 0: kernel %.4f ms" % (i, d.data_ptr(), run(srcs[0], d)), flush=True)
for i, s in enumerate(srcs):
    print("src %d (%#x), dst 0: kernel %.4f ms" % (i, s.data_ptr(), run(s, dsts[0])), flush=True)
for lz in (8, 16, 32, 64):
    op.set_q2_chunk(lz)
    print("z-chunk %d: kernel %.4f ms" % (lz, run(srcs[0], dsts[0])), flush=True)

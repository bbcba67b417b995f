#!/usr/bin/env python3
"""development probe: does the time of the Q2/Q1 vmult at 128^3 depend on where src / dst sit relative to the
state array (HBM channel mapping)?  Same engine, same state; dst (and src) are views into one large
torch buffer at different byte offsets; kernel-only time from the engine's event timer."""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import adaflo_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
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
big = torch.zeros(nu * 3 + (1 << 24), dtype=torch.float64, device="cuda")
su = torch.from_numpy(rng.uniform(-1, 1, nu)).cuda()
sp = torch.from_numpy(rng.uniform(-1, 1, npp)).cuda()
dp = torch.empty(npp, dtype=torch.float64, device="cuda")
src_p, dst_p = op.wrap(sp), op.wrap(dp)


def run(off_dst, off_src, reps=20):
    d = big[off_dst // 8: off_dst // 8 + nu]
    if off_src is None:
        s = su
    else:
        s = big[nu + (1 << 21) + off_src // 8: nu + (1 << 21) + off_src // 8 + nu]
        s.copy_(su)
    src = adaflo_amd.BlockVector([op.wrap(s), src_p])
    dst = adaflo_amd.BlockVector([op.wrap(d), dst_p])
    for _ in range(5):
        op.vmult(dst, src)
    op.synchronize()
    op.get_kernel_statistics()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        op.vmult(dst, src)
    e1.record()
    torch.cuda.synchronize()
    ksec, kcount = op.get_kernel_statistics()
    return e0.elapsed_time(e1) / reps, 1e3 * ksec / max(kcount, 1)


for rnd in range(2):
    for off in (0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, 1 << 21, 3 << 20, 1 << 23):
        ms, kms = run(off, None)
        print("round %d dst offset %9d: vmult %.4f ms, kernel %.4f ms" % (rnd, off, ms, kms), flush=True)
for off in (0, 4096, 1 << 20, 1 << 21):
    ms, kms = run(0, off)
    print("src offset %9d: vmult %.4f ms, kernel %.4f ms" % (off, ms, kms), flush=True)

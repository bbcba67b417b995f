"""Timing probe (round 6, review item 3): can the seam fix-up run UNDER the FP64-bound main kernel?  Two contexts on two
streams: A runs the main kernel of a whole vmult (phase 3: all workgroups, no fix-up), B the fix-up pass alone (phase 4) --
on data of its own, so the numbers mean nothing, only the times do.  Prints: main alone, fix-up alone, both launched
back to back on their streams, and main + fix-up of ONE context in sequence (what a vmult does today)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import numpy as np
import adaflo_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
fp = adaflo_amd.FlowParameters(velocity_degree=2)
ts = adaflo_amd.TimeStepping(fp)
for _ in range(3):
    ts.next()
ops = []
for i in range(2):
    op = adaflo_amd.NavierStokesMatrix(fp, adaflo_amd.BrickMesh([n] * 3, [-1] * 3, [1] * 3))
    op.initialize(ts, True)
    rng = np.random.default_rng(i)
    sol = op.block_vector(rng.uniform(-1, 1, op.n_dofs_u()), rng.uniform(-1, 1, op.n_dofs_p()))
    old = adaflo_amd.BlockVector([op.initialize_u_vector(rng.uniform(-1, 1, op.n_dofs_u()))])
    rhs, dst = op.block_vector(), op.block_vector()
    op.residual(rhs, sol, None, old, old)
    op.vmult(dst, sol)                       # (slabs allocated, lists built)
    op.vmult_phase(dst, sol, 3, 0)
    op.vmult_phase(dst, sol, 4, 0)
    op.synchronize()
    ops.append((op, sol, dst))
(a, sa, da), (b, sb, db) = ops


def timeit(f, reps=40):
    for _ in range(5):
        f()
    a.synchronize(); b.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    a.synchronize(); b.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


print("main alone        %.4f ms" % timeit(lambda: a.vmult_phase(da, sa, 3, 0)))
print("fix-up alone      %.4f ms" % timeit(lambda: b.vmult_phase(db, sb, 4, 0)))
print("sequence (1 ctx)  %.4f ms" % timeit(lambda: (a.vmult_phase(da, sa, 3, 0), a.vmult_phase(da, sa, 4, 0))))
print("both, two streams %.4f ms" % timeit(lambda: (a.vmult_phase(da, sa, 3, 0), b.vmult_phase(db, sb, 4, 0))))
print("both, fix-up first %.4f ms" % timeit(lambda: (b.vmult_phase(db, sb, 4, 0), a.vmult_phase(da, sa, 3, 0))))
print("whole vmult       %.4f ms" % timeit(lambda: a.vmult(da, sa)))

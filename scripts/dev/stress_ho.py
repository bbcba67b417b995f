#!/usr/bin/env python3
"""Hunt for run-to-run differences of the high-order sweep kernel: fresh engine contexts on poisoned device memory,
repeated launches, bitwise comparison with the first result; prints which DoFs differ.
usage: stress_ho.py k nx ny nz [repetitions]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import adaflo_amd  # noqa: E402
from common import Case  # noqa: E402


def main():
    k, ncell = int(sys.argv[1]), tuple(int(a) for a in sys.argv[2:5])
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 40
    case = Case(ncell, k=k, upper=(1.0, 0.5, 2.0), tau_grad_div=0.2)
    src_u, src_p, lin = case.random_u(), case.random_p(), case.random_lin()
    first = None
    nn = [k * n + 1 for n in ncell]
    for r in range(reps):
        # poison BEFORE the engine allocates anything (tables, projection weights, partial-sum buffers)
        tmp = case.engine()
        tctx = tmp._require()
        junk0 = [adaflo_amd.DeviceVector.from_numpy(tctx, np.full(1 << (7 + i % 14), np.nan if r % 2 else 1e300)) for i in range(800)]
        del junk0, tmp
        op = case.engine()
        ctx = op._require()
        # poison: memory the engine allocates afterwards (slabs, state copies) starts as NaN or as huge numbers
        junk = [adaflo_amd.DeviceVector.from_numpy(ctx, np.full(1 << 22, np.nan if r % 2 else 1e300)) for _ in range(3)]
        junk += [adaflo_amd.DeviceVector.from_numpy(ctx, np.full(1 << (7 + i % 12), np.nan if r % 2 else 1e300)) for i in range(600)]
        del junk
        op.set_kernel_variant(2)
        op.set_linearization(lin)
        src = op.block_vector(src_u, src_p)
        for inner in range(3):
            dst = op.block_vector(np.full(case.n_u, 7.0), np.full(case.n_p, -3.0))
            op.vmult(dst, src)
            gu, gp = dst.numpy()
            if first is None:
                first = (gu.copy(), gp.copy())
                continue
            du, dp = np.flatnonzero(gu != first[0]), np.flatnonzero(gp != first[1])
            if du.size or dp.size or not np.isfinite(gu).all():
                print("rep %d launch %d: %d velocity / %d pressure entries differ" % (r, inner, du.size, dp.size))
                for e in du[:12]:
                    node, comp = divmod(int(e), 3)
                    i, j, kk = node % nn[0], (node // nn[0]) % nn[1], node // (nn[0] * nn[1])
                    print("   u node (%d, %d, %d) comp %d: %r vs %r" % (i, j, kk, comp, gu[e], first[0][e]))
                npn = [(k - 1) * n + 1 for n in ncell]
                for e in dp[:12]:
                    i, j, kk = e % npn[0], (e // npn[0]) % npn[1], e // (npn[0] * npn[1])
                    print("   p node (%d, %d, %d): %r vs %r" % (i, j, kk, gp[e], first[1][e]))
        del op
    print("done: %d repetitions" % reps)


if __name__ == "__main__":
    main()

import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "scripts")); sys.path.insert(0, ROOT)
import bench_ops as b
for lin in (0, 1, 2, 4, 3):
    b.ns_case(2, 128, 1, state_from_residual=True, linearization=lin)
for v in (1, 0):                                 # round 6: explicit scheme with variable coefficients on the sweep kernel / generic
    b.ns_case(2, 128, v, two_phase=True, linearization=3)
for lin in (2, 3):                               # round 6: two-phase residual of the schemes that linearise about the extrapolated velocity
    for v in (1, 0):
        b.ns_residual_case(2, 128, v, two_phase=True, linearization=lin)

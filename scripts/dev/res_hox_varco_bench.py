import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "scripts")); sys.path.insert(0, ROOT)
import bench_ops as b
for k, n in ((3, 64), (4, 64)):
    for v in (1, 0):
        b.ns_residual_case(k, n, v, two_phase=True)
for k, n in ((2, 128), (4, 64)):
    for v in (1, 0):
        b.ns_residual_case(k, n, v, linearization=4)

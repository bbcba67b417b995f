import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "scripts")); sys.path.insert(0, ROOT)
import bench_ops as b
for lazy in (True, False):
    b.ns_residual_case(2, 128, 1, lazy=lazy)
    b.ns_residual_case(2, 128, 1, two_phase=True, lazy=lazy)

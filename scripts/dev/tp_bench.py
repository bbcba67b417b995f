import sys
sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
import bench_ops as b
for v in (1, 4):
    b.ns_case(2, 128, v, two_phase=True, state_from_residual=True)

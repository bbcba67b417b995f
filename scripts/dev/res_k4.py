import sys
sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
import bench_ops as b
b.ns_residual_case(4, 64, 1)
b.ns_residual_case(4, 64, 1)
b.ns_residual_case(3, 64, 1)
b.ns_residual_case(5, 48, 1)

import sys, json
sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
import bench_ops as b
b.ns_residual_case(2, 128, 1)
b.ns_residual_case(2, 128, 1, two_phase=True)   # sweep kernel with the coefficients from the generic arrays (round 5)
b.ns_residual_case(2, 128, 0, two_phase=True)   # generic kernel
b.ns_residual_case(4, 64, 1)

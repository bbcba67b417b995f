#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
timeout 600 python -m pytest -q -m gpu -x tests/test_ns_parity_gpu.py -k "test_residual_x_marching_kernel" 2>&1 | tail -30
timeout 1200 python -m pytest -q -m gpu -x tests/test_lb_differential_gpu.py 2>&1 | tail -40

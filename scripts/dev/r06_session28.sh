#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
timeout 2400 python -m pytest -q -m gpu -x tests/test_ns_parity_gpu.py tests/test_state_machine_gpu.py tests/test_lb_differential_gpu.py 2>&1 | tail -6
for i in 1 2 3; do timeout 300 python tests/probe_residual.py 5,3,2,3,2 5,2,3,3,2 5,1,1,3,2 2>&1 | grep "rep 0\|fault" | cut -c1-110; done
python scripts/dev/res_ext_bench.py 2>&1 | grep "^{" | cut -c1-200 | tail -2

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
timeout 2400 python -m pytest -q -m gpu -x tests/test_ns_parity_gpu.py tests/test_state_machine_gpu.py tests/test_lb_differential_gpu.py tests/test_full_size_gpu.py tests/test_two_phase_gpu.py 2>&1 | grep -E "passed|failed" | tail -3
bash scripts/dev/pmc_res_lazy.sh 2>&1 | tail -4
python scripts/dev/res_lazy_bench.py 2>&1 | grep "^{" | cut -c1-160

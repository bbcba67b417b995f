#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06
mkdir -p $O
timeout 3000 python -m pytest -q -m gpu tests/test_state_machine_gpu.py tests/test_ns_parity_gpu.py tests/test_two_phase_gpu.py tests/test_navier_stokes_gpu.py tests/test_boundary_gpu.py 2>&1 | grep -E "^E   +AssertionError|passed|failed" | cut -c1-700 | tee $O/fail_sm.log

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
timeout 2400 python -m pytest -q -m gpu -x tests/test_ns_parity_gpu.py -k "two_phase_residual_x_marching or random_meshes" 2>&1 | tail -15
timeout 1200 python -m pytest -q -m gpu -x tests/test_state_machine_gpu.py 2>&1 | tail -5
python scripts/dev/res_hox_varco_bench.py 2>&1 | grep "^{" | cut -c1-250

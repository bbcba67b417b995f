#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
python scripts/dev/res_lazy_bench.py 2>&1 | tail -6
timeout 2400 python -m pytest -q -m gpu -x tests/test_state_machine_gpu.py tests/test_ns_parity_gpu.py tests/test_full_size_gpu.py 2>&1 | tail -15
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
timeout 2400 python -m pytest -q -m gpu -x tests/test_ns_parity_gpu.py -k "residual" 2>&1 | tail -15

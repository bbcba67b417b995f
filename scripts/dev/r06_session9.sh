#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06
mkdir -p $O
timeout 4000 python -m pytest tests -q -m gpu --maxfail=10 --deselect tests/test_full_size_gpu.py::test_config3_256cubed_on_one_gpu_properties 2>&1 | tail -40 > $O/pytest_gpu_b.log; tail -40 $O/pytest_gpu_b.log

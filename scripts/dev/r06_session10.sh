#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06
mkdir -p $O
timeout 3000 python -m pytest -q -m gpu -x "tests/test_full_size_gpu.py::test_timed_path_small_meshes_with_the_naive_oracle_velocity_block" 2>&1 | grep -v "^$" | tail -40 | cut -c1-400 > $O/fail_timed.log
timeout 3000 python -m pytest -q -m gpu -s tests/test_lb_differential_gpu.py 2>&1 | grep -E "^E|passed|failed|bitwise" | cut -c1-1500 | head -30 > $O/fail_lbd.log
timeout 3000 python -m pytest -q -m gpu tests/test_state_machine_gpu.py 2>&1 | grep -E "^E   +AssertionError|passed|failed" | cut -c1-700 > $O/fail_sm.log
cat $O/fail_timed.log $O/fail_lbd.log $O/fail_sm.log
timeout 900 python bench.py --gpus 2 --cells 32 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_dev
mkdir -p $O
for i in 1 2 3; do
python bench.py --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('128^3', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
timeout 2000 python -m pytest -q -m gpu -x tests/test_ns_parity_gpu.py tests/test_two_phase_gpu.py tests/test_lb_differential_gpu.py tests/test_golden_gpu.py "tests/test_full_size_gpu.py::test_timed_path_config2_128cubed_residual_then_recomputed_vmult_against_openmp_oracle" tests/test_parallel_gpu.py 2>&1 | tail -4

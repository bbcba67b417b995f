#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r06
timeout 3600 python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/r06/pytest_gpu.log
tail -5 gpurun_out/r06/pytest_gpu.log
bash scripts/measure_round6.sh r06
tail -1 gpurun_out/r06/bench_n1_plain.log | cut -c1-400
tail -1 gpurun_out/r06/bench_cavity_q4_plain.log | cut -c1-300

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
for i in 1 2; do
python bench.py --config cavity --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cavity 64^3', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06_dev/cav -o cav -- python3 $GRAFT_REPO_ROOT/bench.py --config cavity --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python3 scripts/kstats.py gpurun_out/r06_dev/cav 3
python scripts/bench_ho.py 2>/dev/null | tail -8 | cut -c1-200
timeout 2000 python -m pytest -q -m gpu tests/test_ns_parity_gpu.py tests/test_full_size_gpu.py tests/test_lb_differential_gpu.py 2>&1 | grep -E "passed|failed" | tail -3

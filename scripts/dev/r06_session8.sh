#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06
mkdir -p $O
echo "=== old tree b79a1e7 with the kernel header of b0746b5 (node lines through one per-lane address), k = 5, HOX_EXT_LB=1" | tee $O/k5_old_b0746.log
(cd _wt_old && for i in 1 2; do timeout 300 python scripts/dev/res_k5_probe.py 5,1,1,1,2 5,3,2,3,2 2>&1 | grep -v "^ " | tail -8 | tee -a ../$O/k5_old_b0746.log; done)
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest_gpu_a.log
python bench.py > $O/bench_n1_a.log 2>&1; tail -1 $O/bench_n1_a.log | cut -c1-1500
python bench.py --config cavity --no-cpu-baseline > $O/bench_cavity_a.log 2>&1; tail -1 $O/bench_cavity_a.log | cut -c1-600

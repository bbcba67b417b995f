#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
export ADAFLO_LIB_PATH=$PWD/adaflo_amd/lib/variants/lib_hox_prealloc.so
for i in 1 2 3 4 5; do
timeout 300 python tests/probe_residual.py 5,3,2,3,2 5,3,2,3,4 2>&1 | grep "rep 0\|fault"
done
python scripts/dev/res_ext_bench.py 2>&1 | grep "^{" | cut -c1-200 | tail -6
python bench.py --config cavity --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cavity 64^3 prealloc', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
unset ADAFLO_LIB_PATH
python bench.py --config cavity --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cavity 64^3 product', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"

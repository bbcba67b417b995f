#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06
mkdir -p $O
for c in 32 16 8 4 16 11 22 8 32; do
python bench.py --no-cpu-baseline --steps 100 --warmup 10 --chunk $c 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('128^3 chunk $c', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done 2>&1 | tee $O/chunk_sweep.log
for c in 32 16 8 20; do
python bench.py --no-cpu-baseline --steps 30 --warmup 5 --cells 160 --chunk $c 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('160^3 chunk $c', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done 2>&1 | tee -a $O/chunk_sweep.log
for c in 32 16; do
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --cells 256 --chunk $c 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('256^3 chunk $c', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done 2>&1 | tee -a $O/chunk_sweep.log

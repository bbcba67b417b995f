#!/bin/bash
# development: Q4/Q3 with the whole-cell state ring at one workgroup per CU (lib_hox_deep.so = build_variant.sh hox_deep
# ns_hox -DHOX_DEEP=1) against the product build (two-point ring, two workgroups per CU); cycles per phase of both and of
# the product kernel at one workgroup per CU (-DHOX_LB=1 -DHOX_STAMP=1)
cd "$(dirname "$0")/../.."
V=adaflo_amd/lib/variants
for rep in 1 2; do
  echo "== product"; python bench.py --config cavity --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1
  echo "== deep ring"; ADAFLO_LIB_PATH=$V/lib_hox_deep.so python bench.py --config cavity --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1
done
echo "== deep ring, stamps"; ADAFLO_LIB_PATH=$V/lib_hox_deeps.so python bench.py --config cavity --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | grep "hox stamp"
echo "== LB=1, stamps, Newton"; ADAFLO_LIB_PATH=$V/lib_hox_lb1s.so python bench.py --config cavity --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | grep "hox stamp"
echo "== LB=1, stamps, Stokes"; ADAFLO_LIB_PATH=$V/lib_hox_lb1s.so python scripts/bench_ho.py 4 1 2>&1 | grep "hox stamp\|stokes"
ADAFLO_LIB_PATH=$V/lib_hox_deep.so python -m pytest tests/test_ns_parity_gpu.py -m gpu -x -q -k "x_marching or high_order or q4 or hox" 2>&1 | tail -3

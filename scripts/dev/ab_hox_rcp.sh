#!/bin/bash
# development: recompute-state mode of the x-marching kernel (variant 1) at two / one workgroup per CU against the
# streamed state (variant 4).  Build first (the product library does not instantiate the mode):
#   scripts/dev/build_variant.sh hox_rcp2 ns_hox -DHOX_RCP_BUILD=1
#   scripts/dev/build_variant.sh hox_rcp1 ns_hox -DHOX_RCP_BUILD=1 -DHOX_RCP_LB=1      (+ -DHOX_STAMP=1: cycles per phase)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for rep in 1 2; do
    echo "== recompute LB=2"; ADAFLO_LIB_PATH=adaflo_amd/lib/variants/lib_hox_rcp2.so python bench.py --config cavity --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1
    echo "== recompute LB=1"; ADAFLO_LIB_PATH=adaflo_amd/lib/variants/lib_hox_rcp1.so python bench.py --config cavity --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1
    echo "== streamed"; python bench.py --config cavity --variant 4 --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1
done
for k in 3 5; do
  echo "== k=$k recompute LB=2 / LB=1 / streamed"
  ADAFLO_LIB_PATH=adaflo_amd/lib/variants/lib_hox_rcp2.so python bench.py --config cavity --degree $k --cells $((k==3?64:48)) --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1
  ADAFLO_LIB_PATH=adaflo_amd/lib/variants/lib_hox_rcp1.so python bench.py --config cavity --degree $k --cells $((k==3?64:48)) --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1
  python bench.py --config cavity --degree $k --cells $((k==3?64:48)) --variant 4 --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1
done
python -m pytest tests/test_ns_parity_gpu.py tests/test_full_size_gpu.py tests/test_navier_stokes_gpu.py -m gpu -x -q 2>&1 | tail -5

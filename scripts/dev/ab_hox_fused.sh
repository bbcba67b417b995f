#!/bin/bash
# development: A/B of the x-marching kernel with value+gradient carried through two exchanges (HOX_FUSED=1, product
# build) against the round-4 form (lib_hox_old.so = scripts/dev/build_variant.sh hox_old ns_hox -DHOX_FUSED=0)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for rep in 1 2; do
  for cfg in cavity; do
    echo "== fused $cfg"; python bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1
    echo "== old   $cfg"; ADAFLO_LIB_PATH=adaflo_amd/lib/variants/lib_hox_old.so python bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1
  done
done
python -m pytest tests/test_ns_parity_gpu.py tests/test_full_size_gpu.py -m gpu -x -q 2>&1 | tail -5

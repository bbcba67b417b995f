#!/bin/bash
# development: quad broadcasts of the Q2/Q1 recompute kernel through the LDS crossbar (ds_swizzle_b32) instead of DPP moves
#   scripts/dev/build_variant.sh q2_sw<mask> ns_q2 -DQ2_SWIZZLE=<mask>   (bit 0: gradient matrix, 1: u / p, 2: u_lin, 3: div u_lin)
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  for v in "" 1 14 15; do
    l=""; [ -n "$v" ] && l=adaflo_amd/lib/variants/lib_q2_sw$v.so
    echo "== mask ${v:-0}"; ADAFLO_LIB_PATH=$l python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
done

#!/bin/bash
# development: how the wrong 512-register build of the Q2/Q1 extrapolating residual was bisected: build the unit
# twice with the flags given (-DQ2_EXT_LB=1 / =2 added; during the bisection: switches that turned parts of the new code off),
# run one residual with each on the GPU box, compare bitwise
#   usage (here): scripts/dev/lb_diff_ext.sh build <tag> <flags...>      then     gpurun -- scripts/dev/lb_diff_ext.sh run <tag>
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  tag=$2; shift 2
  scripts/dev/build_variant.sh ext1$tag ns_q2 -DQ2_EXT_LB=1 "$@" > /dev/null 2>&1 &
  scripts/dev/build_variant.sh ext2$tag ns_q2 -DQ2_EXT_LB=2 "$@" > /dev/null 2>&1 &
  wait; ls adaflo_amd/lib/variants/lib_ext1$tag.so adaflo_amd/lib/variants/lib_ext2$tag.so
else
  tag=$2
  mkdir -p gpurun_out
  for lin in "coupled velocity explicit" "coupled velocity semi-implicit"; do
    for v in 1 2; do ADAFLO_LIB_PATH=adaflo_amd/lib/variants/lib_ext$v$tag.so python scripts/dev/lb_diff_one.py /tmp/e$v.npy "$lin" 8 8 4 2>/dev/null; done
    python -c "
import numpy as np
a,b=np.load('/tmp/e1.npy'),np.load('/tmp/e2.npy')
print('$tag $lin: max abs diff', np.abs(a-b).max(), 'of', np.abs(b).max(), 'entries differing', int((a!=b).sum()), 'of', a.size)"
  done
fi

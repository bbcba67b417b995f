#!/bin/bash
# round 6, GPU session 1: hazard probe; the two register-allocation-dependent wrong results against variant builds; Q4/Q3 counters
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06
mkdir -p $O
hipcc --offload-arch=gfx950 -O2 scripts/dev/hazard_probe.hip -o /tmp/hazard_probe > $O/hazard_probe_build.log 2>&1
timeout 300 /tmp/hazard_probe > $O/hazard_probe.log 2>&1
echo "hazard probe rc $?"; cat $O/hazard_probe.log
# Q2/Q1 extrapolating residual: one residual per build, bitwise against the 256-register build
for lin in "coupled velocity explicit" "coupled velocity semi-implicit"; do
  for v in q2ext2 q2ext1ng q2ext1pad; do
    ADAFLO_LIB_PATH=adaflo_amd/lib/variants/lib_$v.so timeout 300 python scripts/dev/lb_diff_one.py /tmp/$v.npy "$lin" 8 8 4 2>/dev/null
  done
  timeout 300 python scripts/dev/lb_diff_one.py /tmp/product.npy "$lin" 8 8 4 2>/dev/null
  python - "$lin" <<'PY' 2>&1 | tee -a $O/q2_ext_diff.log
import sys, numpy as np
b = np.load('/tmp/q2ext2.npy')
for v in ('q2ext1ng', 'q2ext1pad', 'product'):
    a = np.load('/tmp/%s.npy' % v)
    print('%-32s %-10s max abs diff %.3e of %.3e, entries differing %d of %d' % (sys.argv[1], v, np.abs(a - b).max(), np.abs(b).max(), int((a != b).sum()), a.size))
PY
done
# Q5/Q4 extrapolating residual at one workgroup per CU
for v in hx0 hx1 hx2 hx3 hx4; do
  echo "=== $v" | tee -a $O/k5_ext.log
  ADAFLO_LIB_PATH=adaflo_amd/lib/variants/lib_$v.so timeout 600 python tests/probe_residual.py 5,1,1,1,2 5,3,2,3,2 5,3,2,3,3 2>&1 | grep -v "^ " | tee -a $O/k5_ext.log
done
# Q4/Q3 counters (Newton and, for comparison, the kernel without a state stream)
bash scripts/dev/pmc_hox.sh r06 > $O/pmc_hox.log 2>&1
tail -30 $O/pmc_hox.log

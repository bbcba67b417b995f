#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06
mkdir -p $O
lin="coupled velocity explicit"
for v in q2ext2 g1 g2 g4 g8; do
  ADAFLO_LIB_PATH=adaflo_amd/lib/variants/lib_$v.so timeout 300 python scripts/dev/lb_diff_one.py /tmp/$v.npy "$lin" 8 8 4 2>/dev/null
done
python - <<'PY' 2>&1 | tee $O/q2_ext_sites.log
import numpy as np
b = np.load('/tmp/q2ext2.npy')
for v, n in (('g1', 'guard on the quadrature-loop broadcasts only'), ('g2', 'guard on the broadcasts of the extrapolated field only'),
             ('g4', 'guard on the phase-E broadcasts only'), ('g8', 'no guard (coefficient sites do not exist in this kernel)')):
    a = np.load('/tmp/%s.npy' % v)
    print('%-70s max abs diff %.3e, entries differing %d of %d' % (n, np.abs(a - b).max(), int((a != b).sum()), a.size))
PY

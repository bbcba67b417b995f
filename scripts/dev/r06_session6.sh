#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06
mkdir -p $O
lin="coupled velocity explicit"
for v in q2ext2 n0 n1 n2 n3; do
  ADAFLO_LIB_PATH=adaflo_amd/lib/variants/lib_$v.so timeout 300 python scripts/dev/lb_diff_one.py /tmp/$v.npy "$lin" 8 8 4 2>&1 | tail -2
done
python - <<'PY' 2>&1 | tee $O/q2_ext_nops.log
import numpy as np
b = np.load('/tmp/q2ext2.npy')
for v, n in (('n0', 'wrong build, code object re-assembled from its unchanged listing'), ('n1', '... s_nop 7 behind EVERY VALU instruction of the kernel'),
             ('n2', '... s_nop 7 ahead of and behind every DPP move'), ('n3', '... s_nop 7 around v_accvgpr_*, v_readlane, v_writelane')):
    try:
        a = np.load('/tmp/%s.npy' % v)
    except OSError:
        print('%-70s no result' % n); continue
    print('%-70s max abs diff %.3e, entries differing %d of %d' % (n, np.abs(a - b).max(), int((a != b).sum()), a.size))
PY

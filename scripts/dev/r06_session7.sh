#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06
mkdir -p $O
for v in q2ext2 n0 p1 p2; do
  ADAFLO_LIB_PATH=adaflo_amd/lib/variants/lib_$v.so timeout 300 python scripts/dev/poison_probe.py /tmp/pp_$v.npy 2>/dev/null
done
python - <<'PY' 2>&1 | tee $O/q2_ext_poison.log
import numpy as np
np.set_printoptions(linewidth=220, precision=4, suppress=True)
ref = np.load('/tmp/pp_q2ext2.npy')
nu = 17 * 17 * 9 * 3
for v in ('n0', 'p1', 'p2'):
    a = np.load('/tmp/pp_%s.npy' % v)
    print('==', v, '(n0: wrong build; p1: wrong build + poison; p2: guarded build + poison)  max |diff| per case', np.abs(a - ref).max(axis=1))
    for case in range(3):
        d = (a[case, :nu] - ref[case, :nu]).reshape(9, 17, 17, 3)
        vals, cnt = np.unique(np.round(d[np.abs(d) > 1e-9], 5), return_counts=True)
        order = np.argsort(-cnt)[:8]
        print('   case u_%d = x_%d: distinct differences (value: count):' % (case, case), ', '.join('%.5f: %d' % (vals[i], cnt[i]) for i in order))
        for val in vals[order][:4]:
            n = (abs(val) / 2 - 1) * 256
            print('        %.5f -> register number if it is a poison value: %.2f' % (val, n))
PY

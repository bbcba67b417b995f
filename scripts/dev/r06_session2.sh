#!/bin/bash
# round 6, GPU session 2: compiler-flag bisect of the wrong 512-register build (Q2/Q1 extrapolating residual, unguarded);
# where its wrong entries sit; the k = 5 residual of the commit that first showed wrong pressure rows; bench supervisor tests
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06
mkdir -p $O
lin="coupled velocity explicit"
for v in q2ext2 q2ext1ng f1 f2 f3 f4 f5 f6 f7; do
  ADAFLO_LIB_PATH=adaflo_amd/lib/variants/lib_$v.so timeout 300 python scripts/dev/lb_diff_one.py /tmp/$v.npy "$lin" 8 8 4 2>/dev/null
done
python - <<'PY' 2>&1 | tee $O/q2_ext_flags.log
import numpy as np
b = np.load('/tmp/q2ext2.npy')
names = {'q2ext1ng': 'unguarded, 512 registers', 'f1': '-amdgpu-dpp-combine=0', 'f2': '-enable-post-misched=0', 'f3': '-enable-misched=0',
         'f4': '-amdgpu-waitcnt-forcezero=1', 'f5': '-amdgpu-spill-vgpr-to-agpr=0', 'f6': '-vgpr-regalloc=basic',
         'f7': '-amdgpu-sdwa-peephole=0 -amdgpu-enable-rewrite-partial-reg-uses=0'}
for v, n in names.items():
    try:
        a = np.load('/tmp/%s.npy' % v)
    except OSError:
        print('%-70s no result' % n); continue
    print('%-70s max abs diff %.3e, entries differing %d of %d' % (n, np.abs(a - b).max(), int((a != b).sum()), a.size))
a = np.load('/tmp/q2ext1ng.npy')
nu = 17 * 17 * 9 * 3
du = (a[:nu] != b[:nu]).reshape(9, 17, 17, 3)
dp = (a[nu:] != b[nu:]).reshape(5, 9, 9)
print('velocity rows differing per component:', du.sum(axis=(0, 1, 2)), ' pressure rows:', dp.sum())
print('per z-plane (u):', du.sum(axis=(1, 2, 3)), ' (p):', dp.sum(axis=(1, 2)))
print('per y-line  (u):', du.sum(axis=(0, 2, 3)))
print('per x-column(u):', du.sum(axis=(0, 1, 3)))
PY
echo "=== old tree (b79a1e7 + k = 5 enabled, HOX_EXT_LB=1)" | tee $O/k5_old.log
(cd _wt_old && timeout 600 python scripts/dev/res_k5_probe.py 5,1,1,1,2 5,3,2,3,2 2>&1 | grep -v "^ " | tee -a ../$O/k5_old.log)
timeout 900 python -m pytest tests/test_bench_contract.py -x -q -m gpu 2>&1 | tail -5 | tee $O/bench_contract.log

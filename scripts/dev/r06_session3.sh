#!/bin/bash
# round 6, GPU session 3: which term of the extrapolating residual is wrong in the unguarded 512-register build; two more
# compiler flags; the k = 5 extrapolating residual at one workgroup per CU through the parity tests
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06
mkdir -p $O
for v in q2ext2 q2ext1ng; do
  ADAFLO_LIB_PATH=adaflo_amd/lib/variants/lib_$v.so timeout 300 python scripts/dev/ext_term_probe.py /tmp/t_$v.npz 2>/dev/null
done
python scripts/dev/ext_term_probe.py --compare /tmp/t_q2ext1ng.npz /tmp/t_q2ext2.npz 2>&1 | tee $O/q2_ext_terms.log
lin="coupled velocity explicit"
for v in q2ext2 f8 f9; do
  ADAFLO_LIB_PATH=adaflo_amd/lib/variants/lib_$v.so timeout 300 python scripts/dev/lb_diff_one.py /tmp/$v.npy "$lin" 8 8 4 2>/dev/null
done
python - <<'PY' 2>&1 | tee -a $O/q2_ext_flags.log
import numpy as np
b = np.load('/tmp/q2ext2.npy')
for v, n in (('f8', '-amdgpu-sdwa-peephole=0'), ('f9', '-amdgpu-enable-rewrite-partial-reg-uses=0')):
    a = np.load('/tmp/%s.npy' % v)
    print('%-70s max abs diff %.3e, entries differing %d of %d' % (n, np.abs(a - b).max(), int((a != b).sum()), a.size))
PY
echo "=== k = 5 extrapolating residual at one workgroup per CU (variant library hx0): parity tests" | tee $O/k5_ext_tests.log
ADAFLO_LIB_PATH=adaflo_amd/lib/variants/lib_hx0.so timeout 1500 python -m pytest tests/test_ns_parity_gpu.py -x -q -m gpu -k "residual" 2>&1 | tail -5 | tee -a $O/k5_ext_tests.log
ADAFLO_LIB_PATH=adaflo_amd/lib/variants/lib_hx0.so timeout 600 python scripts/dev/res_ext_bench.py 2>&1 | tail -12 | tee -a $O/k5_ext_tests.log
echo "=== old tree, k = 4 then k = 5" | tee $O/k5_old.log
(cd _wt_old && timeout 300 python scripts/dev/res_k5_probe.py 4,3,2,3,2 2>&1 | grep -v "^ " | tail -3 | tee -a ../$O/k5_old.log; timeout 300 python scripts/dev/res_k5_probe.py 5,1,1,1,2 2>&1 | grep -v "^ " | tail -6 | tee -a ../$O/k5_old.log)

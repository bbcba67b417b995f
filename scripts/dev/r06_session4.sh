#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06
mkdir -p $O
for v in q2ext2 q2ext1ng; do
  ADAFLO_LIB_PATH=adaflo_amd/lib/variants/lib_$v.so timeout 300 python scripts/dev/ext_term_probe.py /tmp/t_$v.npz 2>/dev/null
done
python scripts/dev/ext_term_probe.py --compare /tmp/t_q2ext1ng.npz /tmp/t_q2ext2.npz > $O/q2_ext_inputs.log 2>&1

#!/bin/bash
# development: the 512-register kernels against 256-register builds of the same source, bitwise (see lb_differential.py)
#   scripts/dev/build_variant.sh q2_lb2 ns_q2 -DQ2_RES_LB=2 -DQ2_RCP_LB=2
#   scripts/dev/build_variant.sh hox_lb2 ns_hox -DHOX_RES_LB=2 -DHOX_EXT_LB=2      (two libraries: one unit each)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python scripts/dev/lb_differential.py /tmp/lbd_product.npz
ADAFLO_LIB_PATH=adaflo_amd/lib/variants/lib_q2_lb2.so python scripts/dev/lb_differential.py /tmp/lbd_q2_lb2.npz
ADAFLO_LIB_PATH=adaflo_amd/lib/variants/lib_hox_lb2.so python scripts/dev/lb_differential.py /tmp/lbd_hox_lb2.npz
python - <<'PY'
import numpy as np
a = np.load("/tmp/lbd_product.npz")
for name in ("q2_lb2", "hox_lb2"):
    b = np.load("/tmp/lbd_%s.npz" % name)
    worst = 0.0
    for key in a.files:
        x, y = a[key], b[key]
        d = np.abs(x - y).max() / max(np.abs(x).max(), 1e-300)
        if d > 0:
            print("  %-10s %-70s max rel diff %.2e" % (name, key, d))
        worst = max(worst, d)
    print(name, "worst relative difference over", len(a.files), "arrays:", worst)
PY

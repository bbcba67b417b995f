#!/bin/bash
# level-set sweep kernel experiments: rebuild csrc/q1_sweep.hip with the given defines and time the nodal operators
# usage (GPU box): bash scripts/dev/exp_q1.sh "" "-DQ1_EXP_NOLOAD"
cd ${GRAFT_REPO_ROOT:-.}
for e in "$@"; do
  hipcc -c adaflo_amd/csrc/q1_sweep.hip -o adaflo_amd/lib/q1_sweep.o -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-gpu-rdc $e || continue
  hipcc -shared -o adaflo_amd/lib/libadaflo_hip.so adaflo_amd/lib/*.o --offload-arch=gfx950 -fno-gpu-rdc
  echo "exp [$e]"
  python scripts/dev/adv_bench.py 2>&1 | grep "64, 64, 128" | grep "vmult" | cut -c1-110
done

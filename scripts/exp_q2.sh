set -e
cd $GRAFT_REPO_ROOT
# usage: exp_q2.sh "<defines>" ...   e.g. "-DQ2_EXP=4 -DQ2_RING=18"
for e in "$@"; do
  hipcc -c adaflo_amd/csrc/ns_q2.hip -o adaflo_amd/lib/ns_q2.o -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-gpu-rdc $e
  hipcc -shared -o adaflo_amd/lib/libadaflo_hip.so adaflo_amd/lib/*.o --offload-arch=gfx950 -fno-gpu-rdc
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('exp [$e]', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done

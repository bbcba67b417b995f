set -e
cd $GRAFT_REPO_ROOT
# usage: exp_ho.sh "<defines>" ...   development aid for ns_ho.hip, e.g. "-DHO_EXP=1"
for e in "$@"; do
  hipcc -c adaflo_amd/csrc/ns_ho.hip -o adaflo_amd/lib/ns_ho.o -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-gpu-rdc $e
  hipcc -shared -o adaflo_amd/lib/libadaflo_hip.so adaflo_amd/lib/*.o --offload-arch=gfx950 -fno-gpu-rdc
  echo "exp [$e]"
  python scripts/bench_ho.py ${HO_K:-4} 2>&1 | grep -v "variant\": 0"
done

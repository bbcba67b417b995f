set -e
cd $GRAFT_REPO_ROOT
# usage: exp_ho.sh "<defines>" ...   development aid for ns_hox.hip / ns_ho.hip, e.g. "-DHOX_LB=3"
# environment: HO_K (degrees, default 4), HO_V (variants, default 1), HO_LX (x-chunks, default 0 = heuristic)
for e in "$@"; do
  hipcc -c adaflo_amd/csrc/ns_hox.hip -o adaflo_amd/lib/ns_hox.o -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-gpu-rdc $e
  hipcc -c adaflo_amd/csrc/ns_ho.hip -o adaflo_amd/lib/ns_ho.o -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-gpu-rdc $e
  hipcc -shared -o adaflo_amd/lib/libadaflo_hip.so adaflo_amd/lib/*.o --offload-arch=gfx950 -fno-gpu-rdc
  echo "exp [$e]"
  python scripts/bench_ho.py ${HO_K:-4} ${HO_V:-1} ${HO_LX:-0} 2>&1
done

#!/bin/bash
# development: build the engine with extra flags for ONE translation unit into a library of its own (the product
# library adaflo_amd/lib/libadaflo_hip.so is not touched); select it with ADAFLO_LIB_PATH
#   usage: scripts/dev/build_variant.sh <tag> <unit, e.g. ns_hop> <flags...>   ->  gpurun_out/lib_<tag>.so (travels? no:
#   gpurun_out is not pushed) -> adaflo_amd/lib/variants/lib_<tag>.so
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
tag=$1; unit=$2; shift 2
mkdir -p $R/adaflo_amd/lib/variants
hipcc -c $R/adaflo_amd/csrc/$unit.hip -o $R/adaflo_amd/lib/variants/${unit}_$tag.o -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-gpu-rdc "$@"
objs=$(ls $R/adaflo_amd/lib/*.o | grep -v "/$unit.o")
hipcc -shared -o $R/adaflo_amd/lib/variants/lib_$tag.so $objs $R/adaflo_amd/lib/variants/${unit}_$tag.o --offload-arch=gfx950 -fno-gpu-rdc
echo $R/adaflo_amd/lib/variants/lib_$tag.so

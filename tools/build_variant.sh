#!/bin/bash
# development: a second build of the library with extra compiler flags, for A/B runs on the GPU box
#   tools/build_variant.sh <name> <file.hip> <flags...>   ->  nvdiffrast_amd/libnvdr_hip_<name>.so  (use with NVDR_LIB_PATH)
set -e
cd "$(dirname "$0")/../nvdiffrast_amd"
name=$1; src=$2; shift 2
# (the other objects are taken from nvdiffrast_amd/build/ as they are: build the main library first, from the sources you mean)
ls build/*.hip.o >/dev/null
mkdir -p build/ab_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function "$@" -c csrc/$src -o build/ab_$name/$src.o
objs=""
for f in build/*.hip.o; do b=$(basename $f); if [ "$b" == "$src.o" ]; then objs="$objs build/ab_$name/$src.o"; else objs="$objs $f"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o libnvdr_hip_$name.so
echo nvdiffrast_amd/libnvdr_hip_$name.so

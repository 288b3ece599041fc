#!/bin/bash
# development: A/B builds of one source file for a single gpurun call
#   tools/build_ab.sh <file.hip> <name>=<flags or @git-rev> ...
# each variant -> nvdiffrast_amd/libnvdr_hip_<name>.so (select with NVDR_LIB_PATH); "@rev" compiles the file as it was at that commit
set -e
cd "$(dirname "$0")/../nvdiffrast_amd"
src=$1; shift
ls build/*.hip.o >/dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -I../include -Icsrc"
case "$src" in raster.hip|backward_fused.hip|texture.hip) FLAGS="$FLAGS -fno-slp-vectorize";; esac      # (nvdiffrast_amd/_build.py EXTRA_FLAGS)
for spec in "$@"; do
  name=${spec%%=*}; arg=${spec#*=}
  mkdir -p build/ab_$name
  if [[ "$arg" == @* ]]; then
    rev=${arg#@}
    mkdir -p build/ab_$name/src; for f in csrc/*.hpp; do git show $rev:nvdiffrast_amd/$f > build/ab_$name/src/$(basename $f); done
    git show $rev:nvdiffrast_amd/csrc/$src > build/ab_$name/src/$src
    /opt/rocm/bin/hipcc $FLAGS -Ibuild/ab_$name/src -c build/ab_$name/src/$src -o build/ab_$name/$src.o &
  else
    /opt/rocm/bin/hipcc $FLAGS $arg -c csrc/$src -o build/ab_$name/$src.o &
  fi
done
wait
for spec in "$@"; do
  name=${spec%%=*}
  objs=""
  for f in build/*.hip.o; do b=$(basename $f); if [ "$b" == "$src.o" ]; then objs="$objs build/ab_$name/$src.o"; else objs="$objs $f"; fi; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o libnvdr_hip_$name.so
  echo nvdiffrast_amd/libnvdr_hip_$name.so
done

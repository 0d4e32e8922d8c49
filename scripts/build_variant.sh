#!/bin/bash
# Developer helper: builds a variant of libpie_hip.so with ONE source recompiled under extra flags (ablation builds).
#   scripts/build_variant.sh <name> <source.hip> <extra hipcc flags...>   ->  tools/variants/libpie_<name>.so  (git-ignored, travels with gpurun)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p tools/variants/obj
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -D__HIP_PLATFORM_AMD__"
/opt/rocm/bin/hipcc $FLAGS "$@" -c proxy_inference_engine_amd/csrc/$src -o tools/variants/obj/$name.o
objs=""
for o in proxy_inference_engine_amd/lib/obj/*.o; do
  case "$o" in *"/$src.o") objs="$objs tools/variants/obj/$name.o";; *) objs="$objs $o";; esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/variants/libpie_$name.so $objs -ldl
echo tools/variants/libpie_$name.so

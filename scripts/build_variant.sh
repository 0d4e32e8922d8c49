#!/bin/bash
# Developer helper: builds a variant of libpie_hip.so with ONE source recompiled under extra flags (ablation builds).
#   scripts/build_variant.sh <name> <source.hip> <extra hipcc flags...>   ->  tools/variants/<name>/libpie_hip.so  (git-ignored, travels with gpurun)
# and tools/variants/libpie_<name>.so, a link to it: LD_LIBRARY_PATH=tools/variants/<name> (scripts/ab_variants.sh, tools/step_bench) and
# PIE_HIP_LIB=tools/variants/libpie_<name>.so (scripts/ab_batch.sh, the Python host) select the same file.
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
mkdir -p tools/variants/$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/variants/$name/libpie_hip.so $objs -ldl
ln -sf $name/libpie_hip.so tools/variants/libpie_$name.so
echo tools/variants/$name/libpie_hip.so

#!/bin/bash
# Developer helper: a variant of libpie_hip.so with SOME sources recompiled under extra flags.
#   scripts/build_flag_variant.sh <name> "<extra hipcc flags>" <source> [<source> ...]   ->  tools/variants/<name>/libpie_hip.so
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2; shift 2
mkdir -p tools/variants/obj tools/variants/$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -D__HIP_PLATFORM_AMD__"
objs=""
for o in proxy_inference_engine_amd/lib/obj/*.o; do
  b=$(basename $o .o); hit=0
  for src in "$@"; do [ "$b" = "$src" ] && hit=1; done
  if [ $hit = 1 ]; then
    extra=""; [ "$b" = "decoder.hip" ] && extra="-DPIE_BUILD_HASH=\"$name\""
    /opt/rocm/bin/hipcc $FLAGS $flags $extra -c proxy_inference_engine_amd/csrc/$b -o tools/variants/obj/$name.$b.o
    objs="$objs tools/variants/obj/$name.$b.o"
  else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/variants/$name/libpie_hip.so $objs -ldl
echo tools/variants/$name/libpie_hip.so

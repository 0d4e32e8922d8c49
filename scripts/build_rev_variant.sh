#!/bin/bash
# Developer helper: builds libpie_hip.so from the sources of a git revision -> tools/variants/<name>/libpie_hip.so (A/B baseline on the same box).
#   scripts/build_rev_variant.sh <name> <git-rev>
set -e
cd "$(dirname "$0")/.."
name=$1; rev=$2
tmp=$(mktemp -d)
git archive "$rev" proxy_inference_engine_amd/csrc include | tar -x -C "$tmp"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -D__HIP_PLATFORM_AMD__"
mkdir -p tools/variants/$name "$tmp/obj"
pids=""
for src in "$tmp"/proxy_inference_engine_amd/csrc/*.hip "$tmp"/proxy_inference_engine_amd/csrc/*.cpp; do
  ( /opt/rocm/bin/hipcc $FLAGS -DPIE_BUILD_HASH="\"$rev\"" -c "$src" -o "$tmp/obj/$(basename "$src").o" ) &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/variants/$name/libpie_hip.so "$tmp"/obj/*.o -ldl
ln -sf $name/libpie_hip.so tools/variants/libpie_$name.so
rm -rf "$tmp"
echo tools/variants/$name/libpie_hip.so

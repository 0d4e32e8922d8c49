#!/bin/bash
# Developer helper: scripts/bench_w4l.py over library variants (tools/variants/<name>/libpie_hip.so; "base" = the product library).
#   scripts/ab_w4l.sh "<variants>" [M]
M=${2:-4096}
for v in $1; do
  lib=tools/variants/$v/libpie_hip.so; [ "$v" = base ] && lib=proxy_inference_engine_amd/lib/libpie_hip.so
  echo "== $v"
  PIE_HIP_LIB=$PWD/$lib timeout -k 10 200 python scripts/bench_w4l.py $M 2>&1 | grep TFLOP
done

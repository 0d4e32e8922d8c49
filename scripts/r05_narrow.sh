#!/bin/bash
# Round 5 (second session): the native narrow units -- the W2S / W6S tests, then the 2- / 6-bit decode step (tools/step_bench).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5b
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_decode.py -m gpu -q -k "int2 or int6 or narrow or int8_g64 or g32_gemv" > $O/narrow_tests.log 2>&1
echo "pytest rc=$?"; tail -n 12 $O/narrow_tests.log
cd /tmp && export TMPDIR=/tmp
B=$R/tools/step_bench
for b in 2 6 4; do
timeout -k 10 200 $B --model 8b --bits $b --steps 256 --warmup 32 > $O/step_bits$b${TAG}.log 2>&1; tail -n 1 $O/step_bits$b${TAG}.log
done

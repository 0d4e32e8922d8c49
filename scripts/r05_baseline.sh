#!/bin/bash
# Round 5, first GPU call: conversion probe + where the prompt path and the 32-sequence step stand before the new GEMM.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5/base
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 $R/tools/dequant_probe > $O/dequant_probe.txt 2>&1
B=$R/tools/step_bench
for M in 32 64 128 256 512; do
  timeout -k 10 120 $B --model 8b --prefill $M --prefill-reps 8 > $O/prefill_$M.log 2>&1
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s128 -- $B --model 8b --prefill 128 --prefill-reps 8 > $O/s128.log 2>&1
cd $R && timeout -k 10 400 python scripts/bench_batch.py --batches 8,32 --steps 32 > $O/batch.log 2>&1
find $O -name "*.csv" -size +20M -delete
cat $O/dequant_probe.txt; for M in 32 64 128 256 512; do tail -n 2 $O/prefill_$M.log; done; tail -n 4 $O/batch.log

#!/bin/bash
# Round 5: the vision tower (Qwen2.5-VL-7B geometry) on the hand-written 16-bit GEMM: timings and per-kernel durations (1024 and 4096 patches).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5/prof_vision
mkdir -p $O
cd $R
PY=/usr/bin/python3.10
$PY scripts/bench_vision.py --grid 16x16 32x32 64x64 > $O/plain.log 2>&1
export TMPDIR=/tmp
for g in 32x32 64x64; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$g -- $PY scripts/bench_vision.py --grid $g --iters 3 > $O/stats_$g.log 2>&1
  f=$(ls -t $(find $O/stats_$g -name "*kernel_stats.csv") | head -1); cp $f $O/vision_${g}_kernel_stats.csv
done
find $O -name "*trace.csv" -delete
cat $O/plain.log | tail -n 4; head -12 $O/vision_64x64_kernel_stats.csv | cut -c1-160

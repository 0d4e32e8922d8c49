#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5/w4r
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/abl.txt
for a in "" _1 _2 _4 _8 _16 _3 _12 _31; do
  timeout -k 10 120 $R/tools/w4r_bench$a time 32 128 2>&1 | grep -E "gate_up|down|qkv" | sed 's/   old.*//' >> $O/abl.txt
done
cat $O/abl.txt

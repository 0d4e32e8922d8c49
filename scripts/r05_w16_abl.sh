#!/bin/bash
# Round 5: ablation builds of the 16-bit GEMM (tools/w16_bench_<mask>, -DW16L_ABL=<mask>) on chosen rows; W16_ONLY filters shapes by substring
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5/w16
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for a in ${ABLS:-1 4 8 5 13 15}; do
  timeout -k 10 200 $R/tools/w16_bench_$a time ${MS:-4096} 2>&1 | grep -E "${W16_ONLY:- (gate_up|down|qkv|t_qkv) }" | tee -a $O/abl.txt
done

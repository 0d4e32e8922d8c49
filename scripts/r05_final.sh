#!/bin/bash
# Round 5, final library: the GPU suite, the decode-step profile passes (kernel stats + PMC traffic: profiles/r05_traffic.json names this build), the bench lines.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5b
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_final.txt 2>&1
echo "pytest rc=$?"; tail -n 4 $O/pytest_gpu_final.txt
bash scripts/profile_r05.sh > $O/prof4.log 2>&1; tail -n 4 $O/prof4.log
cd $R && bash scripts/bench_lines_r05.sh > $O/lines.log 2>&1; tail -n 12 $O/lines.log

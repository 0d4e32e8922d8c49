#!/bin/bash
# Round 5, final library: profile passes (4-, 2-, 6-bit step: kernel stats + PMC traffic), the GPU suite, then the 8B / 2-bit / 6-bit bench lines with the traffic of THIS build.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5b
mkdir -p $O
cd $R
bash scripts/profile_r05.sh > $O/prof4.log 2>&1; tail -n 2 $O/prof4.log
BITS=2 bash scripts/profile_r05.sh > $O/prof2.log 2>&1; tail -n 1 $O/prof2.log
BITS=6 bash scripts/profile_r05.sh > $O/prof6.log 2>&1; tail -n 1 $O/prof6.log
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_final.txt 2>&1
echo "pytest rc=$?"; tail -n 3 $O/pytest_gpu_final.txt

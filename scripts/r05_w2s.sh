#!/bin/bash
# Round 5 (second session): the native 2-bit units (W2S) -- full GPU suite, the 2-bit decode step against the 4-bit one (tools/step_bench), the bench line.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5b
mkdir -p $O
cd $R
timeout -k 10 ${TEST_TIMEOUT:-900} python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -n 6 $O/pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp
B=$R/tools/step_bench
timeout -k 10 200 $B --model 8b --steps 256 --warmup 32 > $O/step_bits4.log 2>&1; tail -n 3 $O/step_bits4.log
timeout -k 10 200 $B --model 8b --bits 2 --steps 256 --warmup 32 > $O/step_bits2.log 2>&1; tail -n 3 $O/step_bits2.log
cd $R && timeout -k 10 500 python bench.py --bits 2 > $O/r05_bench_line_bits2.json 2> $O/bits2.err; tail -c 1500 $O/r05_bench_line_bits2.json; tail -n 3 $O/bits2.err

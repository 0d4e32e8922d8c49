#!/bin/bash
# Round 5: the GPU suite + the prompt / multi-sequence timings on the library as built.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5/t${TAG:-0}
mkdir -p $O
cd $R
timeout -k 10 ${TEST_TIMEOUT:-900} python -m pytest tests -m gpu -x -q ${PYTEST_ARGS} > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -n 15 $O/pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp
B=$R/tools/step_bench
for M in 8 32 64 128 256 512; do
  timeout -k 10 120 $B --model 8b --prefill $M --prefill-reps 8 > $O/prefill_$M.log 2>&1; tail -n 1 $O/prefill_$M.log
done
cd $R && timeout -k 10 400 python scripts/bench_batch.py --batches 8,16,32 --steps 32 > $O/batch.log 2>&1; tail -n 4 $O/batch.log

#!/bin/bash
# Round 5: per-kernel durations of the prompt path (128 / 512 tokens) and of the 32-sequence step (rocprofv3 --kernel-trace --stats).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5/prof${TAG:-}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$R/tools/step_bench
for M in ${PROMPTS:-128 512}; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$M -- $B --model 8b --prefill $M --prefill-reps 8 > $O/s$M.log 2>&1
  f=$(find $O/s$M -name "*kernel_stats.csv" | head -1); cp $f $O/prefill${M}_kernel_stats.csv 2>/dev/null
done
if [ -z "$NO_BATCH" ]; then
cd $R && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b32 -- /usr/bin/python3.10 scripts/bench_batch.py --batches 32 --steps 32 > $O/b32.log 2>&1
f=$(find $O/b32 -name "*kernel_stats.csv" | head -1); cp $f $O/batch32_kernel_stats.csv 2>/dev/null
fi
find $O -name "*trace.csv" -delete; find $O -name "*.csv" -size +20M -delete
for M in ${PROMPTS:-128 512}; do echo "== prefill $M"; head -14 $O/prefill${M}_kernel_stats.csv | cut -c1-150; done
[ -z "$NO_BATCH" ] && { echo "== batch 32"; head -24 $O/batch32_kernel_stats.csv | cut -c1-150; }

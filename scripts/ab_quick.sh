#!/bin/bash
# quick same-box A/B through tools/step_bench: scripts/ab_quick.sh [reps] -- r3 library vs the working tree's, split-KV and per-head attention plans
R=${GRAFT_REPO_ROOT:-$(pwd)}
A="--model 8b --mode launch --steps 256 --warmup 32 --no-mega"
for rep in $(seq 1 ${1:-3}); do
  echo "r3      heads 0: $(LD_LIBRARY_PATH=$R/tools/variants/r3 $R/tools/step_bench $A --heads 0 | tail -1)"
  echo "current heads 0: $($R/tools/step_bench $A --heads 0 | tail -1)"
  echo "r3      heads 1: $(LD_LIBRARY_PATH=$R/tools/variants/r3 $R/tools/step_bench $A --heads 1 | tail -1)"
  echo "current heads 1: $($R/tools/step_bench $A --heads 1 | tail -1)"
done

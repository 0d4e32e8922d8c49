#!/bin/bash
# same-box A/B of library variants (tools/variants/<name>) against the working tree's library: scripts/ab_variants4.sh "r3 pin pre_tp" [reps]
R=${GRAFT_REPO_ROOT:-$(pwd)}
A="--model 8b --steps 256 --warmup 32"
for rep in $(seq 1 ${2:-3}); do
  for v in $1; do echo "$v: $(LD_LIBRARY_PATH=$R/tools/variants/$v $R/tools/step_bench $A | tail -1)"; done
  echo "current: $($R/tools/step_bench $A | tail -1)"
done

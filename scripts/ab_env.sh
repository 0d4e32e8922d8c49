#!/bin/bash
# A/B of environment switches through tools/step_bench (developer helper): scripts/ab_env.sh "NAME=VAL ..." "NAME=VAL ..." ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
B=$R/tools/step_bench
ARGS=${STEP_ARGS:---model 8b --mode launch --no-mega --steps 1024 --warmup 64}
for rep in 1 2 3; do
  for e in "$@"; do
    out=$(env $e $B $ARGS 2>&1 | grep "launch sequence")
    echo "[$e] $out"
  done
done

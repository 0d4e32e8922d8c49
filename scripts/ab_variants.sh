#!/bin/bash
# A/B of library variants through tools/step_bench (developer helper; variants under tools/variants/<name>/libpie_hip.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}
B=$R/tools/step_bench
run() { # name, env...
  local name=$1; shift
  local out
  out=$(env "$@" $B ${STEP_ARGS:---model 8b --mode launch --no-mega --steps 1024 --warmup 64} 2>&1 | grep "launch sequence")
  echo "$name: $out"
}
for rep in 1 2 3; do
  run base X=1
  for v in "$@"; do run $v LD_LIBRARY_PATH=$R/tools/variants/$v; done
  run nopf PIE_PREFETCH_MB=-1
done

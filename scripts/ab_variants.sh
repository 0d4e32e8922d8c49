#!/bin/bash
# A/B of library variants through tools/step_bench (developer helper; variants under tools/variants/<name>/libpie_hip.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}
B=$R/tools/step_bench
run() { # name, env...
  local name=$1; shift
  local out
  out=$(env "$@" $B ${STEP_ARGS:---model 8b --mode launch --no-mega --steps 1024 --warmup 64} 2>&1 | grep -E "launch sequence|persistent step|library")
  echo "$name: $out"
}
for rep in 1 2 3; do
  run base X=1
  for v in "$@"; do
    [ -f $R/tools/variants/$v/libpie_hip.so ] || { echo "no such variant: tools/variants/$v/libpie_hip.so (scripts/build_variant.sh $v ...)"; exit 1; }
    run $v LD_LIBRARY_PATH=$R/tools/variants/$v   # step_bench prints the library it loaded
  done

done

#!/bin/bash
# Round 5: same-box A/B of the decode step through tools/step_bench: tools/variants/$1 (scripts/build_rev_variant.sh) vs the working tree's library, alternating
R=${GRAFT_REPO_ROOT:-$(pwd)}
V=${1:-prev}
A="--model ${MODEL:-8b} --steps 256 --warmup 32 ${EXTRA}"
cd /tmp
for rep in 1 2 3; do
  echo "$V      : $(LD_LIBRARY_PATH=$R/tools/variants/$V timeout -k 10 120 $R/tools/step_bench $A | tail -1)"
  echo "current : $(timeout -k 10 120 $R/tools/step_bench $A | tail -1)"
done

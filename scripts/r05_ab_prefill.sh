#!/bin/bash
# Round 5: same-box A/B of the prompt path through tools/step_bench: tools/variants/$1 (scripts/build_rev_variant.sh) vs the working tree's library, alternating
R=${GRAFT_REPO_ROOT:-$(pwd)}
V=${1:-prev}
cd /tmp
for rep in 1 2 3; do
  for M in ${PROMPTS:-256 512 1024 4096}; do
    reps=6; [ $M -ge 2048 ] && reps=3
    a=$(LD_LIBRARY_PATH=$R/tools/variants/$V timeout -k 10 200 $R/tools/step_bench --model 8b --prefill $M --prefill-reps $reps | tail -1)
    b=$(timeout -k 10 200 $R/tools/step_bench --model 8b --prefill $M --prefill-reps $reps | tail -1)
    echo "rep $rep  $V: $a   |   current: $b"
  done
done

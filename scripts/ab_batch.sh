R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R
for rep in 1 2; do
  for v in base prio; do
    if [ $v = base ]; then unset PIE_HIP_LIB; else export PIE_HIP_LIB=$R/tools/variants/libpie_$v.so; fi
    echo "[$v] $(python $R/scripts/bench_batch.py --batches 8,32 --steps 48 2>&1 | grep step_batch | tr '\n' ' ')"
  done
done

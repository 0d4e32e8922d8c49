#!/bin/bash
# Developer helper: the decode step under runtime (ROCclr) environment settings that change how graph kernel nodes are dispatched.
R=${GRAFT_REPO_ROOT:-$(pwd)}
B=$R/tools/step_bench
for rep in 1 2; do
  for e in "X=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "AMD_OPT_FLUSH=0" "AMD_OPT_FLUSH=1" \
           "DEBUG_HIP_GRAPH_BATCH_SIZE=1024" "DEBUG_HIP_KERNARG_COPY_OPT=0" "GPU_MAX_HW_QUEUES=2" "DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0" "AMD_DIRECT_DISPATCH=0"; do
    echo "$e: $(env $e timeout -k 5 60 $B --model 8b --mode launch --no-mega --steps 768 --warmup 64 2>&1 | grep 'launch sequence')"
  done
done

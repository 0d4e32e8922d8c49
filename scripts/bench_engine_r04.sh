#!/bin/bash
# Round 4: BatchedEngine end to end on the 8B int4 model (one MI355X): mixed passes, shared prompt prefix, chunked prefill.
for a in "" "--no-mixed" "--prompt 512 --shared 384" "--prompt 512 --shared 384 --share-prefix" "--prompt 2048 --requests 16 --slots 8" "--prompt 2048 --requests 16 --slots 8 --chunk 256" "--prompt 2048 --requests 16 --slots 8 --chunk 512"; do
  echo "== bench_batch.py --engine $a"
  timeout -k 10 400 python scripts/bench_batch.py --engine $a 2>/dev/null | tail -1
done

#!/bin/bash
# Developer sweep: (row tile, K split) of the many-row int4 GEMM at small row counts, 8B shapes (scripts/bench_w4l.py under PIE_W4L_MT / PIE_W4L_S)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for M in ${ROWS:-64 128 256}; do
  echo "== M=$M default plan"; python $R/scripts/bench_w4l.py $M | awk '{print "   ", $1, $5, "ms"}'
  for mt in 64 128 256; do
    [ $mt -gt 64 ] && [ $M -le $((mt / 2)) ] && continue
    for S in 1 2 4 7 8 14 16; do
      echo "== M=$M mt=$mt S=$S"; PIE_W4L_MT=$mt PIE_W4L_S=$S python $R/scripts/bench_w4l.py $M | awk '{print "   ", $1, $5, "ms"}'
    done
  done
done

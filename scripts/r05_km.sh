#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5/w4r
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in 1 2; do
W4R_CFG=$c timeout -k 10 300 $R/tools/w4r_bench_km check > $O/check_km$c.txt 2>&1; echo "km cfg $c: $(grep -c 'BAD 0$' $O/check_km$c.txt) ok; $(tail -n 1 $O/check_km$c.txt)"
echo "== K-major, cfg $c"; W4R_CFG=$c timeout -k 10 300 $R/tools/w4r_bench_km time 8 32 64 128 256 2>&1 | sed 's/   old.*//'
done
echo "== K-major, no conversion, cfg 2"; W4R_CFG=2 timeout -k 10 300 $R/tools/w4r_bench_km1 time 32 128 2>&1 | sed 's/   old.*//'

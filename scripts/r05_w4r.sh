#!/bin/bash
# Round 5: the weight-streaming GEMM alone (tools/w4r_bench) -- parity grid with both conversions, timing against the kernels it replaces.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5/w4r
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in 0 1; do
  W4R_PLAIN=$c timeout -k 10 300 $R/tools/w4r_bench check > $O/check_plain$c.txt 2>&1
  rc=$?
  echo "plain=$c: $(grep -c 'BAD 0$' $O/check_plain$c.txt) shapes ok, $(grep -c 'BAD [1-9]' $O/check_plain$c.txt) bad"; grep -v "BAD 0$" $O/check_plain$c.txt | tail -n 12
  [ $rc -ne 0 ] && { echo "check rc=$rc: timing skipped"; exit 1; }
done
timeout -k 10 300 $R/tools/w4r_bench time ${MS:-8 32 64 128 256} > $O/time.txt 2>&1
cat $O/time.txt

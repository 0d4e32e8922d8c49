#!/bin/bash
# Round 5: the 16-bit many-row GEMM alone (tools/w16_bench) -- parity grid, then timing against hipBLASLt.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5/w16
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 $R/tools/w16_bench check > $O/check.txt 2>&1
rc=$?
echo "$(grep -c 'BAD 0$' $O/check.txt) shapes ok, $(grep -c 'BAD [1-9]' $O/check.txt) bad"; grep -v "BAD 0$" $O/check.txt | tail -n 12
[ $rc -ne 0 ] && { echo "check rc=$rc: timing skipped"; exit 1; }
timeout -k 10 500 $R/tools/w16_bench time ${MS:-512 1024 4096} > $O/time${TAG}.txt 2>&1
cat $O/time${TAG}.txt

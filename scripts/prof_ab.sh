#!/bin/bash
# kernel-trace stats of tools/step_bench for: the r3 library, the current library (pilot off), the current library (pilot N)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4/prof_ab
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$R/tools/step_bench
A="--model 8b --mode launch --no-mega --steps 64 --warmup 8 --sync-every 4"
LD_LIBRARY_PATH=$R/tools/variants/r3 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3 -- $B $A > $O/r3.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/new0 -- $B $A --pilot 0 > $O/new0.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/new1 -- $B $A --pilot ${1:-1} > $O/new1.log 2>&1
for v in r3 new0 new1; do echo "== $v"; tail -1 $O/$v.log; f=$(ls $O/$v/*/*kernel_stats.csv | head -1); cut -d, -f1-4 $f | head -12 | cut -c1-160; done
find $O -name "*trace.csv" -size +8M -delete

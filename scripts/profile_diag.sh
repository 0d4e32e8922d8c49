#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2/diag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$R/tools/step_bench
try() {  # name, command...
    n=$1; shift
    timeout -k 10 200 "$@" > $O/$n.log 2>&1
    rc=$?
    echo "$n rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping"; exit 1; fi
}
try plain $B --model 8b --mode launch --steps 32 --warmup 4 --no-mega
try A_nomega_graph rocprofv3 --kernel-trace --stats --output-format csv -d $O/A -- $B --model 8b --mode launch --steps 32 --warmup 4 --no-mega
try B_nomega_eager rocprofv3 --kernel-trace --stats --output-format csv -d $O/B -- $B --model 8b --mode launch --steps 32 --warmup 4 --no-mega --graph 0
try C_tiny rocprofv3 --kernel-trace --stats --output-format csv -d $O/C -- $B --model tiny --mode launch --steps 8 --warmup 2 --no-mega --graph 0
cd $R
try D_python rocprofv3 --kernel-trace --stats --output-format csv -d $O/D -- python3 bench.py --no-cpu-baseline --steps 16 --warmup 4
find $O -name "*.csv" -size +20M -delete
tail -4 $O/*.log

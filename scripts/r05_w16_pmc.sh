#!/bin/bash
# Round 5: average core clock of the 16-bit GEMM and of hipBLASLt's kernel on one shape (GRBM_GUI_ACTIVE / duration), and their busy cycles
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5/w16/pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export W16_SHAPE=${W16_SHAPE:-gate_up}
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES -d $O -o pmc --output-format csv -- $R/tools/w16_bench time ${MS:-4096} > $O/run.log 2>&1
tail -n 3 $O/run.log
ls -R $O | head -20

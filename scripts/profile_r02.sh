#!/bin/bash
# Round-2 profiling passes on the MI355X box (run through gpurun from the repo root):
#   kernel trace of the product decode step and of a 4096-token prefill through the C-ABI driver tools/step_bench,
#   then PMC passes (separate runs, --pmc only) for HBM traffic of the step and MFMA activity of the prompt GEMMs.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$R/tools/step_bench
# un-profiled first touch: on a fresh box the very first profiled process segfaulted inside rocprofv3's dispatch interception
# (twice, before any kernel of the step ran); after one plain run of the same binary every profiled run below went through
$B --model 8b --mode launch --no-mega --steps 32 --warmup 4 > $O/plain.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/step_stats -- $B --model 8b --mode launch --no-mega --steps 64 --warmup 8 --sync-every 4 > $O/step_stats.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prefill_stats -- $B --model 8b --no-mega --prefill 4096 --prefill-reps 3 > $O/prefill_stats.log 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/step_fetch -- $B --model 8b --mode launch --no-mega --graph 0 --steps 8 --warmup 2 --sync-every 1 > $O/step_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/step_write -- $B --model 8b --mode launch --no-mega --graph 0 --steps 8 --warmup 2 --sync-every 1 > $O/step_write.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES --output-format csv -d $O/prefill_mfma -- $B --model 8b --no-mega --prefill 4096 --prefill-reps 1 > $O/prefill_mfma.log 2>&1 || echo "mfma pass failed" >> $O/prefill_mfma.log
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/prefill_busy -- $B --model 8b --no-mega --prefill 4096 --prefill-reps 1 > $O/prefill_busy.log 2>&1 || echo "busy pass failed" >> $O/prefill_busy.log
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES --output-format csv -d $O/prefill16_mfma -- $B --model 8b --no-mega --prefill 16 --prefill-reps 8 > $O/prefill16_mfma.log 2>&1 || echo "mfma16 pass failed" >> $O/prefill16_mfma.log
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prefill16_stats -- $B --model 8b --no-mega --prefill 16 --prefill-reps 8 > $O/prefill16_stats.log 2>&1 || echo "stats16 pass failed" >> $O/prefill16_stats.log
(cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python3 bench.py --no-cpu-baseline --steps 64 --warmup 8 > $O/bench_stats.log 2>&1) || echo "bench stats pass failed" >> $O/bench_stats.log
(rocprofv3 -L 2>/dev/null | grep -i "mfma\|SQ_BUSY\|FETCH_SIZE\|WRITE_SIZE" | head -60) > $O/counters.txt || true
# keep only the csv summaries small enough to merge back
find $O -name "*.csv" -size +30M -delete
du -sh $O
for f in $O/*.log; do echo "== $f"; tail -n 3 $f; done

#!/bin/bash
# Round 4: the prompt path's MFMA activity (4096-token prefill of the 8B int4 model through pie_decoder_prefill): kernel trace + two PMC passes
# (separate runs, --pmc only).  Run through gpurun from the repo root; scripts/summarize_r04_prefill.py turns the output into profiles/r04_prefill_mfma.json.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4/prof_prefill
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$R/tools/step_bench
run() { local name=$1; shift; timeout -k 10 300 "$@" > $O/$name.log 2>&1 || echo "$name FAILED ($?)" | tee -a $O/$name.log; }
$B --model 8b --prefill 4096 --prefill-reps 3 > $O/plain.log 2>&1 || true
run prefill_stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/prefill_stats -- $B --model 8b --prefill 4096 --prefill-reps 3
run prefill_mfma rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES --output-format csv -d $O/prefill_mfma -- $B --model 8b --prefill 4096 --prefill-reps 1
run prefill_busy rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/prefill_busy -- $B --model 8b --prefill 4096 --prefill-reps 1
/usr/bin/python3.10 -c "import sys; sys.path.insert(0, '$R'); from proxy_inference_engine_amd import _ffi; print(_ffi.load().pie_version().decode())" > $O/pie_version.txt 2>/dev/null || true
find $O -name "*trace.csv" -size +16M -delete
find $O -name "*.csv" -size +30M -delete
for f in $O/*.log; do echo "== $f"; tail -n 2 $f | cut -c1-200; done

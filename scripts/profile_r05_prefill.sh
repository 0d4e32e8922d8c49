#!/bin/bash
# Round 5: the prompt path under rocprofv3 on the FINAL library -- per-kernel durations (--kernel-trace --stats) and MFMA activity (two --pmc
# passes, separate runs) for the int4 8B model at 128 / 512 / 4096 tokens (tools/step_bench) and for the dense bf16 model at 512 tokens
# (configs[2]; scripts/bench_prefill.py --dense under /usr/bin/python3.10).  scripts/summarize_r05_prefill.py condenses the output into profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5/prof_prefill
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$R/tools/step_bench
run() { local name=$1; shift; timeout -k 10 400 "$@" > $O/$name.log 2>&1 || echo "$name FAILED ($?)" | tee -a $O/$name.log; }
for M in ${PROMPTS:-128 512 4096}; do
  reps=8; [ $M -ge 2048 ] && reps=3
  $B --model 8b --prefill $M --prefill-reps $reps > $O/plain_$M.log 2>&1 || true
  run stats_$M rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$M -- $B --model 8b --prefill $M --prefill-reps $reps
  run mfma_$M rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES --output-format csv -d $O/mfma_$M -- $B --model 8b --prefill $M --prefill-reps 1
  run busy_$M rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/busy_$M -- $B --model 8b --prefill $M --prefill-reps 1
done
if [ -z "$NO_DENSE" ]; then
  cd $R
  PY=/usr/bin/python3.10
  $PY scripts/bench_prefill.py --dense --prompts 512 --iterated-max 0 > $O/plain_dense512.log 2>&1 || true
  run stats_dense512 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dense512 -- $PY scripts/bench_prefill.py --dense --prompts 512 --iterated-max 0
  run mfma_dense512 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES --output-format csv -d $O/mfma_dense512 -- $PY scripts/bench_prefill.py --dense --prompts 512 --iterated-max 0
fi
/usr/bin/python3.10 -c "import sys; sys.path.insert(0, '$R'); from proxy_inference_engine_amd import _ffi; print(_ffi.load().pie_version().decode())" > $O/pie_version.txt 2>/dev/null || true
find $O -name "*trace.csv" -size +16M -delete
find $O -name "*.csv" -size +30M -delete
for f in $O/*.log; do echo "== $f"; tail -n 2 $f | cut -c1-200; done

#!/bin/bash
# Round-5 profiling passes on the MI355X box (run through gpurun from the repo root): kernel trace of the product decode step (tools/step_bench
# and bench.py itself), then PMC passes (separate runs, --pmc only) for the step's HBM traffic.  Capture rules: scripts/profile_r03.sh.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
# BITS=2 | 6: the same passes over the 2- / 6-bit step (W2S / W6S units) -> gpurun_out/r5/prof_step_bits$BITS (no bench.py pass)
O=$R/gpurun_out/r5/prof_step${BITS:+_bits$BITS}
BARG=${BITS:+--bits $BITS}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$R/tools/step_bench
run() { local name=$1; shift; timeout -k 10 300 "$@" > $O/$name.log 2>&1 || echo "$name FAILED ($?)" | tee -a $O/$name.log; }
run step_stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/step_stats -- $B --model 8b $BARG --steps 64 --warmup 8 --sync-every 4
run step_fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/step_fetch -- $B --model 8b $BARG --graph 0 --steps 8 --warmup 2 --sync-every 1
run step_write rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/step_write -- $B --model 8b $BARG --graph 0 --steps 8 --warmup 2 --sync-every 1
[ -z "$BITS" ] && (cd $R && run bench_stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- /usr/bin/python3.10 bench.py --no-cpu-baseline --steps 64 --warmup 8)
$B --model 8b $BARG --steps 256 --warmup 32 > $O/plain.log 2>&1 || true
/usr/bin/python3.10 -c "import sys; sys.path.insert(0, '$R'); from proxy_inference_engine_amd import _ffi; print(_ffi.load().pie_version().decode())" > $O/pie_version.txt 2>/dev/null || true
find $O -name "*trace.csv" -size +16M -delete
find $O -name "*.csv" -size +30M -delete
du -sh $O
for f in $O/*.log; do echo "== $f"; tail -n 3 $f | cut -c1-200; done

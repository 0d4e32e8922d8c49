#!/bin/bash
# Round-3 profiling passes on the MI355X box (run through gpurun from the repo root): kernel traces of the product decode step (launch
# sequence, the default) and of the opt-in persistent step, then PMC passes (separate runs, --pmc only) for the step's HBM traffic.
# Capture rules, each with its reason (scripts/profile_diag_r03.sh, profiles/r03_rocprofv3_abort_diagnosis.txt):
#   * the program after `--` is the ELF itself (tools/step_bench, /usr/bin/python3.10): under --pmc the profiler's preloaded library has
#     initialised the GPU before the program starts, so any launcher hop (env, bash -c, a shim) would be an exec after GPU init;
#   * --sync-every bounds the dispatches in flight: with ~25-40 thousand graph-node packets outstanding rocprofv3 7.2 faults inside its
#     interception of hipGraphLaunch (reproduced with a backtrace: C_40k_dispatches.log); a few steps per synchronisation never get there;
#   * no warm-up run: the "first profiled process on a fresh box segfaults" of round 2 did not reproduce (A_first_process_profiled: exit 0).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$R/tools/step_bench
run() { local name=$1; shift; timeout -k 10 240 "$@" > $O/$name.log 2>&1 || echo "$name FAILED ($?)" | tee -a $O/$name.log; }
run step_stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/step_stats -- $B --model 8b --mode launch --no-mega --steps 64 --warmup 8 --sync-every 4
run engine_stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/engine_stats -- $B --model 8b --mode mega --steps 64 --warmup 8 --sync-every 4
run step_fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/step_fetch -- $B --model 8b --mode launch --no-mega --graph 0 --steps 8 --warmup 2 --sync-every 1
run step_write rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/step_write -- $B --model 8b --mode launch --no-mega --graph 0 --steps 8 --warmup 2 --sync-every 1
run engine_fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/engine_fetch -- $B --model 8b --mode mega --graph 0 --steps 8 --warmup 2 --sync-every 1
run engine_write rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/engine_write -- $B --model 8b --mode mega --graph 0 --steps 8 --warmup 2 --sync-every 1
(cd $R && run bench_stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- /usr/bin/python3.10 bench.py --no-cpu-baseline --steps 64 --warmup 8)
$B --model 8b --steps 200 --warmup 20 > $O/plain.log 2>&1 || true
$R/tools/step_bench --model 8b --check 1 --steps 1 --warmup 1 | grep -i "library\|device" > $O/version.log 2>&1 || true
/usr/bin/python3.10 -c "import sys; sys.path.insert(0, '$R'); from proxy_inference_engine_amd import _ffi; print(_ffi.load().pie_version().decode())" > $O/pie_version.txt 2>/dev/null || true
find $O -name "*trace.csv" -size +16M -delete
find $O -name "*.csv" -size +30M -delete
du -sh $O
for f in $O/*.log; do echo "== $f"; tail -n 3 $f | cut -c1-200; done

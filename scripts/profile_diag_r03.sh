#!/bin/bash
# Round-3 diagnosis of the three rocprofv3 aborts recorded in rounds 1-2: each hypothesis gets ONE capture on a fresh box.
#   A  "the first profiled process on a fresh box segfaults": profile tools/step_bench as the very first GPU process, no warm-up run.
#   B  "--pmc + Python host segfaults at the first dispatch": the Python process runs PyTorch's BUNDLED ROCm 7.0 runtime
#      (torch/lib/libamdhip64.so, libhsa-runtime64.so: /proc/self/maps), rocprofv3 is ROCm 7.2's.  B1: pure torch, none of this repo's code,
#      under --pmc; B2: the same with the system 7.2 runtime preloaded into the process.
#   C  "SIGSEGV at ~24k recorded dispatches": 250 graph-replayed steps (40k dispatches) without --sync-every.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3/diag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$R/tools/step_bench
PY=$(readlink -f "$(command -v python3)")
echo "python3 -> $PY ($(file -b $PY | cut -c1-60))" > $O/summary.txt
try() { local name=$1; shift; timeout -k 10 240 "$@" > $O/$name.log 2>&1; echo "$name: exit $?" | tee -a $O/summary.txt; }
try A_first_process_profiled rocprofv3 --kernel-trace --stats --output-format csv -d $O/A -- $B --model 8b --mode launch --no-mega --steps 32 --warmup 4
try A2_pmc_first rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/A2 -- $B --model 8b --mode launch --no-mega --graph 0 --steps 4 --warmup 1 --sync-every 1
try B1_pmc_pure_torch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/B1 -- $PY -c "import torch; x = torch.ones(1 << 20, device='cuda'); print(float((x + 1).sum()))"
export LD_PRELOAD=/opt/rocm/lib/libhsa-runtime64.so.1:/opt/rocm/lib/libamdhip64.so.7
try B2_pmc_torch_system_runtime rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/B2 -- $PY -c "import torch; x = torch.ones(1 << 20, device='cuda'); print(float((x + 1).sum()))"
unset LD_PRELOAD
try C_40k_dispatches rocprofv3 --kernel-trace --stats --output-format csv -d $O/C -- $B --model 8b --mode launch --no-mega --steps 250 --warmup 4 --cap 512
find $O -name "*.csv" -size +8M -delete
for f in $O/*.log; do echo "== $f"; tail -n 4 $f; done >> $O/summary.txt
cat $O/summary.txt | cut -c1-240

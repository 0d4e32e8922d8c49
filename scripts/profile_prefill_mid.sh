set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/p256
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$R/tools/step_bench
$B --model 8b --no-mega --prefill 256 --prefill-reps 8 > $O/plain.log 2>&1
for M in 128 256 512; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$M -- $B --model 8b --no-mega --prefill $M --prefill-reps 8 > $O/s$M.log 2>&1
done
find $O -name "*.csv" -size +20M -delete
tail -n 3 $O/plain.log

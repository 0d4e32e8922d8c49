#!/bin/bash
# Round 5: rehearsal of the driver's N > 1 command on ONE card (two ranks share cuda:0; torch.distributed over gloo): the replicas line plus the
# appended Llama-3-70B tensor-parallel leg (IPC one-shot backend; RCCL refuses two ranks on one device and must be RECORDED as an error, not raised).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5/bench2
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 PIE_BENCH_BACKEND=gloo
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps ${STEPS:-32} --warmup 8 > $O/bench_2rank.log 2>&1
echo "rc=$?"; grep -E '^\{' $O/bench_2rank.log | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(json.dumps({k: d[k] for k in ('value', 'n_gpus', 'ms_per_step', 'tp70b')}, indent=1)); print(json.dumps(d['roofline']['step']['floor'], indent=1))"
tail -n 5 $O/bench_2rank.log | cut -c1-300

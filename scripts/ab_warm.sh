mkdir -p gpurun_out/r4/warm
for cfg in "--dense" "--bits 8" ""; do
 for mb in 0 10 20 1000; do
  python bench.py $cfg --no-cpu-baseline --steps 64 --warmup 8 --knob attn_warm_max_mb=$mb 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg[$cfg] warm_mb=$mb', round(d['repetitions']['median_ms_per_step'],4), d['repetitions']['ms_per_step'])"
 done
done

#!/bin/bash
# Round 5: the bench's JSON line for the four one-card configurations (run on the MI355X box through gpurun) -> gpurun_out/r5/lines/, copied to profiles/ afterwards.
set -e
# (bench.py --model 70b --steps 32 --no-cpu-baseline is appended: configs[4] on one card)
O=gpurun_out/r5/lines; mkdir -p $O
python bench.py > $O/r05_bench_line_8b.json 2> $O/8b.err
python bench.py --dense > $O/r05_bench_line_dense.json 2> $O/dense.err
python bench.py --model qv > $O/r05_bench_line_modelqv.json 2> $O/qv.err
python bench.py --bits 8 > $O/r05_bench_line_bits8.json 2> $O/bits8.err
python bench.py --model 70b --steps 32 --no-cpu-baseline > $O/r05_bench_line_70b.json 2> $O/70b.err
for f in $O/r05_bench_line_*.json; do python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value'],1), round(d['repetitions']['median_ms_per_step'],4), round(d['roofline']['step']['frac'],4), 'traffic' if d['roofline']['traffic'] is not None else 'no-traffic', d.get('parity',{}).get('ok'), d.get('parity',{}).get('ids_checked'))" $f; done
# second session of round 5: the native narrow units
python bench.py --bits 2 > $O/r05_bench_line_bits2.json 2> $O/bits2.err
python bench.py --bits 6 > $O/r05_bench_line_bits6.json 2> $O/bits6.err
for f in $O/r05_bench_line_bits2.json $O/r05_bench_line_bits6.json; do python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value'],1), round(d['repetitions']['median_ms_per_step'],4), round(d['roofline']['step']['frac'],4), d.get('parity',{}).get('ok'), d.get('parity',{}).get('ids_checked'))" $f; done

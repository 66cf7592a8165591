#!/bin/bash
# the round's closing measurements (tools/finals.sh r6: bench line with this run's trace / PMC summaries in place, cfg 3, B = 32) + Large lines
bash tools/finals.sh r6 2>&1 | tail -8
for dt in bf16 fp8; do
  timeout 400 python bench.py --arch large --batch 32 --dtype $dt --steps 20 --warmup 4 --no-cpu-baseline > gpurun_out/r6_large_b32_$dt.json 2> gpurun_out/r6_large_b32_$dt.err < /dev/null
  python -c "import json;d=json.load(open('gpurun_out/r6_large_b32_$dt.json'));print('large B=32 $dt:', d['ms_per_step'], 'ms', d['value'])"
done

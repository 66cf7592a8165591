#!/bin/bash
mkdir -p gpurun_out/r5w
for i in 1 2; do
  for cfg in 512,400 1024,800 768,560 512,288; do
    ST5_WGRAD_ROUND=$cfg timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r5w/r_${cfg}_$i.json 2> gpurun_out/r5w/r_${cfg}_$i.err
    python -c "import json;d=json.load(open('gpurun_out/r5w/r_${cfg}_$i.json'));print('round $cfg run $i', d['ms_per_step'])"
  done
done

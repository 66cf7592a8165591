#!/bin/bash
mkdir -p gpurun_out/r5s
for i in 1 2 3; do
  for g in 0 1; do
    ST5_LAYERDROP_GATE=$g timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r5s/ab_${g}_$i.json 2> gpurun_out/r5s/ab_${g}_$i.err
    python -c "import json;d=json.load(open('gpurun_out/r5s/ab_${g}_$i.json'));print('gate $g run $i', d['ms_per_step'], d['roofline']['frac'])"
  done
done

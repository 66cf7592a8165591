#!/bin/bash
mkdir -p gpurun_out/r5u
timeout 600 python -m pytest tests/test_bf16_path_gpu.py -x -q -m gpu -k "group_launch or tn_glds" > gpurun_out/r5u/t1.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r5u/t1.log | tail -5
timeout 900 python -m pytest tests/test_bench_update_gpu.py -x -q -m gpu  > gpurun_out/r5u/t2.log 2>&1; grep -E "passed|failed|Error" gpurun_out/r5u/t2.log | tail -3
for i in 1 2; do
  for g in 0 1; do
    ST5_WGRAD_GROUP=$g timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r5u/ab_${g}_$i.json 2> gpurun_out/r5u/ab_${g}_$i.err
    python -c "import json;d=json.load(open('gpurun_out/r5u/ab_${g}_$i.json'));print('group $g run $i', d['ms_per_step'], d['roofline']['frac'])"
  done
done

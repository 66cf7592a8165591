#!/bin/bash
mkdir -p gpurun_out/r5t
timeout 1200 python -m pytest tests/test_layerdrop_gpu.py tests/test_bench_update_gpu.py tests/test_cfg2_shape_gpu.py tests/test_two_rank_gpu.py -x -q -m gpu > gpurun_out/r5t/t1.log 2>&1; grep -E "passed|failed" gpurun_out/r5t/t1.log | tail -2
for i in 1 2; do
  timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r5t/new_$i.json 2> gpurun_out/r5t/new_$i.err
  python -c "import json;d=json.load(open('gpurun_out/r5t/new_$i.json'));print('new $i', d['ms_per_step'], d['roofline']['frac'])"
done

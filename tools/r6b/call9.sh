#!/bin/bash
mkdir -p gpurun_out/r6b
timeout 900 python -m pytest tests/test_bf16_path_gpu.py -x -q -k "group or tn" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_bench_update_gpu.py -x -q -k "replayed_equals_eager" 2>&1 | tail -3
for rep in 1 2 3; do
for wm in 16 8; do
  ST5_WG_MAX_P=$wm timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r6b/base_wgmax${wm}_$rep.json 2> gpurun_out/r6b/base_wgmax${wm}_$rep.err < /dev/null
  python -c "import json;d=json.load(open('gpurun_out/r6b/base_wgmax${wm}_$rep.json'));print('base phased group max problems $wm rep $rep:', d['ms_per_step'], 'ms')"
done; done

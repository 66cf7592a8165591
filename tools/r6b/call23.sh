#!/bin/bash
mkdir -p gpurun_out/r6b
timeout 1200 python -m pytest tests/test_graph_gpu.py tests/test_bench_update_gpu.py tests/test_layerdrop_gpu.py -x -q 2>&1 | tail -4 > gpurun_out/r6b/pack_tests_tail.txt; cat gpurun_out/r6b/pack_tests_tail.txt
for rep in 1 2 3; do
for pk in 1 0; do
  ST5_STAGING_PACK=$pk timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r6b/base_pack${pk}_$rep.json 2> gpurun_out/r6b/base_pack${pk}_$rep.err < /dev/null
  python -c "import json;d=json.load(open('gpurun_out/r6b/base_pack${pk}_$rep.json'));print('base staged inputs in one block $pk rep $rep:', d['ms_per_step'], 'ms')"
done; done

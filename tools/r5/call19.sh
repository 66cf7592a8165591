#!/bin/bash
# LayerDrop gate folded into the last LayerNorm: kernel test, select-vs-skip model test, benched-update equality tests, A/B bench
mkdir -p gpurun_out/r5s
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_layerdrop_gpu.py -x -q -m gpu -k "layernorm or select_form" > gpurun_out/r5s/t1.log 2>&1; tail -3 gpurun_out/r5s/t1.log
timeout 900 python -m pytest tests/test_bench_update_gpu.py tests/test_graph_gpu.py -x -q -m gpu > gpurun_out/r5s/t2.log 2>&1; tail -3 gpurun_out/r5s/t2.log
for i in 1 2 3; do
  timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r5s/new_$i.json 2> gpurun_out/r5s/new_$i.err
  python -c "import json;d=json.load(open('gpurun_out/r5s/new_$i.json'));print('new $i', d['ms_per_step'], d['roofline']['frac'])"
done

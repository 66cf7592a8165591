#!/bin/bash
mkdir -p gpurun_out/r6
for rep in 1 2; do
for e in 1 0; do
  ST5_EAGER_TRANSPOSES=$e timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r6/tr_$e_$rep.json 2> gpurun_out/r6/tr_$e_$rep.err
  python -c "import json;d=json.load(open('gpurun_out/r6/tr_$e_$rep.json'));print('transposes refreshed behind Adam = $e, rep $rep:', d['ms_per_step'], 'ms')" || tail -5 gpurun_out/r6/tr_$e_$rep.err
done; done
timeout 1200 python -m pytest tests/test_bench_update_gpu.py tests/test_graph_gpu.py tests/test_flat_optimizer_gpu.py -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -5

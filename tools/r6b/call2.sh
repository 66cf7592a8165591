#!/bin/bash
mkdir -p gpurun_out/r6b
timeout 600 python -m pytest tests/test_fp8_gpu.py -x -q 2>&1 | tail -15
timeout 300 python tools/r6b/fp8_shapes.py > gpurun_out/r6b/fp8_shapes.txt 2>&1; cat gpurun_out/r6b/fp8_shapes.txt
for rep in 1 2; do
for dt in bf16 fp8; do
  timeout 400 python bench.py --arch large --batch 32 --dtype $dt --steps 15 --warmup 4 --no-cpu-baseline > gpurun_out/r6b/large_b32_${dt}_p8_$rep.json 2> gpurun_out/r6b/large_b32_${dt}_p8_$rep.err < /dev/null
  python -c "import json;d=json.load(open('gpurun_out/r6b/large_b32_${dt}_p8_$rep.json'));print('$dt rep $rep:', d['ms_per_step'], 'ms')"
done; done

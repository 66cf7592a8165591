#!/bin/bash
mkdir -p gpurun_out/r6
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k conv0 2>&1 | tail -3
timeout 900 python -m pytest tests/test_two_rank_gpu.py -q -k "without_a_graph" 2>&1 | tail -5
ST5_BENCH_SHAPES=1 timeout 400 python bench.py --steps 20 --no-cpu-baseline > gpurun_out/r6/bench_shapes.json 2> gpurun_out/r6/bench_shapes.err
grep "^#" gpurun_out/r6/bench_shapes.err | head -70

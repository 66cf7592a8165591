#!/bin/bash
O=gpurun_out/r5l; mkdir -p $O
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py tests/test_cfg2_shape_gpu.py tests/test_extractor_ln.py tests/test_bench_update_gpu.py::test_benched_update_replayed_equals_eager_and_reproduces -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2; do timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>$O/err.log | python tools/r5/conv0_line.py; done

#!/bin/bash
mkdir -p gpurun_out/r5v
timeout 300 python tools/r5/group_diff.py 2 2>&1 | grep -E "differ|e-0" | head -10
timeout 1200 python -m pytest tests/test_bench_update_gpu.py tests/test_two_rank_gpu.py tests/test_layerdrop_gpu.py -x -q -m gpu > gpurun_out/r5v/t2.log 2>&1; grep -E "passed|failed|Error" gpurun_out/r5v/t2.log | tail -3

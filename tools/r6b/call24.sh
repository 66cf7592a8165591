#!/bin/bash
mkdir -p gpurun_out/r6b
timeout 1500 python -m pytest tests/test_graph_gpu.py tests/test_bench_update_gpu.py tests/test_layerdrop_gpu.py tests/test_replay_long_gpu.py -x -q > gpurun_out/r6b/pack_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/r6b/pack_tests.log | tail -3

#!/bin/bash
mkdir -p gpurun_out/r6b
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -3 > gpurun_out/r6b/ops_tests_tail.txt; cat gpurun_out/r6b/ops_tests_tail.txt
bash tools/finals.sh r6b 2>&1 | tail -12

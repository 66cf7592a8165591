#!/bin/bash
mkdir -p gpurun_out/r6b
timeout 900 python -m pytest tests/test_flash_gpu.py tests/test_bf16_path_gpu.py -x -q 2>&1 | tail -3 > gpurun_out/r6b/flash_tests_tail.txt; cat gpurun_out/r6b/flash_tests_tail.txt
timeout 300 bash tools/timeline.sh r6b > /dev/null 2>&1; head -3 gpurun_out/timeline_r6b.txt

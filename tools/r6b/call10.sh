#!/bin/bash
mkdir -p gpurun_out/r6b
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r6b/gpu_suite_tail.txt; cat gpurun_out/r6b/gpu_suite_tail.txt

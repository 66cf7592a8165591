#!/bin/bash
mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_bench_update_gpu.py -x -q -k "several_rank or replayed_equals" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -15
timeout 1500 python -m pytest tests/test_two_rank_gpu.py -x -q 2>&1 | tail -15

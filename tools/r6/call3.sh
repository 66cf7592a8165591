#!/bin/bash
mkdir -p gpurun_out/r6
timeout 600 python -m pytest tests/test_bf16_path_gpu.py -q -x -k "half_height or phased_256 or five_slot" 2>&1 | tail -5
echo "== 128^2 always (mode 1)"; ST5_NT_TILE=1 timeout 300 python tools/gemm_cases.py nt 2>&1 | grep "^NT"
echo "== 64x128 always (mode 5)"; ST5_NT_TILE=5 timeout 300 python tools/gemm_cases.py nt 2>&1 | grep "^NT"
echo "== auto"; timeout 300 python tools/gemm_cases.py nt 2>&1 | grep "^NT"

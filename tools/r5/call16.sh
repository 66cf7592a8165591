#!/bin/bash
# bf16 gradient payload: two-rank shared-GPU test + the rest of the two-rank file
mkdir -p gpurun_out/r5p
timeout 1500 python -m pytest tests/test_two_rank_gpu.py -x -q -m gpu -k "bf16_gradient or sharing_the_gpu_equal" > gpurun_out/r5p/two_rank.log 2>&1
tail -5 gpurun_out/r5p/two_rank.log

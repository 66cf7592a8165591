#!/bin/bash
# round 6, call 1: the new cfg-3 parity test, the NT GEMM case table (baseline for this round), the default bench line
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_cfg3_shape_gpu.py -x -q -s 2>&1 | tail -30 > gpurun_out/r6/cfg3_test.log; tail -12 gpurun_out/r6/cfg3_test.log
timeout 300 python tools/gemm_cases.py nt > gpurun_out/r6/gemm_cases_nt_base.txt 2>&1; cat gpurun_out/r6/gemm_cases_nt_base.txt
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r6/bench_base.json 2> gpurun_out/r6/bench_base.err; head -c 400 gpurun_out/r6/bench_base.json

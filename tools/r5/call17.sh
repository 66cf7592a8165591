#!/bin/bash
# which NT launches of the step still run the run-time epilogue form?
mkdir -p gpurun_out/r5q
ST5_GEMM_FEAT_LOG=1 timeout 600 python bench.py --no-graph --steps 1 --warmup 0 > gpurun_out/r5q/line.json 2> gpurun_out/r5q/err.log
grep "rt-epilogue" gpurun_out/r5q/err.log | sort | uniq -c | sort -rn > gpurun_out/r5q/rt_forms.txt
head -40 gpurun_out/r5q/rt_forms.txt

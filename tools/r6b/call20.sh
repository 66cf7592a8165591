#!/bin/bash
mkdir -p gpurun_out/r6b
timeout 900 python -m pytest tests/test_large_gpu.py tests/test_fp8_gpu.py -x -q -k "not oracle" 2>&1 | tail -3 > gpurun_out/r6b/large_tests_tail.txt; cat gpurun_out/r6b/large_tests_tail.txt
for rep in 1 2; do
for km in 1 0; do
  ST5_LNX_DGRAD_KMAJOR=$km timeout 400 python bench.py --arch large --batch 32 --steps 15 --warmup 4 --no-cpu-baseline > gpurun_out/r6b/large_lnx${km}_$rep.json 2> gpurun_out/r6b/large_lnx${km}_$rep.err < /dev/null
  python -c "import json;d=json.load(open('gpurun_out/r6b/large_lnx${km}_$rep.json'));print('large bf16 B=32, K-major extractor dgrad $km rep $rep:', d['ms_per_step'], 'ms')"
done; done

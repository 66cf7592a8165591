#!/bin/bash
mkdir -p gpurun_out/r6b
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_large_gpu.py tests/test_fp8_gpu.py -x -q -k "large or extractor" 2>&1 | tail -4 > gpurun_out/r6b/lnx1p_tests_tail.txt; cat gpurun_out/r6b/lnx1p_tests_tail.txt
for rep in 1 2; do
for fz in 1 0; do
  ST5_LNX_ONEPASS=$fz timeout 400 python bench.py --arch large --batch 32 --steps 15 --warmup 4 --no-cpu-baseline > gpurun_out/r6b/large_lnx1p${fz}_$rep.json 2> gpurun_out/r6b/large_lnx1p${fz}_$rep.err < /dev/null
  python -c "import json;d=json.load(open('gpurun_out/r6b/large_lnx1p${fz}_$rep.json'));print('large bf16 B=32, one-pass pad + tail-only zero fill $fz rep $rep:', d['ms_per_step'], 'ms')"
done; done
ST5_LNX_ONEPASS=1 timeout 400 python bench.py --arch large --batch 32 --dtype fp8 --steps 15 --warmup 4 --no-cpu-baseline > gpurun_out/r6b/large_lnx1p_fp8.json 2> gpurun_out/r6b/large_lnx1p_fp8.err < /dev/null
python -c "import json;d=json.load(open('gpurun_out/r6b/large_lnx1p_fp8.json'));print('large fp8 B=32:', d['ms_per_step'], 'ms')"

#!/bin/bash
mkdir -p gpurun_out/r6b
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_large_gpu.py tests/test_fp8_gpu.py -x -q -k "layernorm or large or fp8 or extractor" > gpurun_out/r6b/relay_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r6b/relay_tests.log | tail -2
for rep in 1 2; do
for rl in 1 0; do
  ST5_PRELN_RELAY=$rl timeout 400 python bench.py --arch large --batch 32 --steps 15 --warmup 4 --no-cpu-baseline > gpurun_out/r6b/large_relay${rl}_$rep.json 2> gpurun_out/r6b/large_relay${rl}_$rep.err < /dev/null
  python -c "import json;d=json.load(open('gpurun_out/r6b/large_relay${rl}_$rep.json'));print('large bf16 B=32, pre-LN residual relay $rl rep $rep:', d['ms_per_step'], 'ms')"
done; done
ST5_PRELN_RELAY=1 timeout 400 python bench.py --arch large --batch 32 --dtype fp8 --steps 15 --warmup 4 --no-cpu-baseline > gpurun_out/r6b/large_relay_fp8.json 2> gpurun_out/r6b/large_relay_fp8.err < /dev/null
python -c "import json;d=json.load(open('gpurun_out/r6b/large_relay_fp8.json'));print('large fp8 B=32:', d['ms_per_step'], 'ms')"

#!/bin/bash
# Vocoder only (tools/bench_cfg3.py, CFG3_VOCODER_ONLY=1), A/B of the two forms of the 32-output-channel convolution kernel.
cd "$(dirname "$0")/.."
for v in 0 1; do
  ST5_NARROW_V1=$v CFG3_VOCODER_ONLY=1 timeout 200 python tools/bench_cfg3.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())['hifigan_forward']
print('ST5_NARROW_V1=$v', {k:d[k] for k in ('ms_per_batch','tflops')})"
done

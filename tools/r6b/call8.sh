#!/bin/bash
mkdir -p gpurun_out/r6b
for rep in 1 2 3; do
for lib in pad nopad; do
  if [ $lib = nopad ]; then export ST5_HIP_LIB=$PWD/speecht5_amd/libspeecht5_hip_nopad.so; else unset ST5_HIP_LIB; fi
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r6b/base_${lib}_$rep.json 2> gpurun_out/r6b/base_${lib}_$rep.err < /dev/null
  python -c "import json;d=json.load(open('gpurun_out/r6b/base_${lib}_$rep.json'));print('base $lib rep $rep:', d['ms_per_step'], 'ms')"
done; done
export ST5_HIP_LIB=$PWD/speecht5_amd/libspeecht5_hip_nopad.so
timeout 900 python -m pytest tests/test_bench_update_gpu.py -x -q -k "replayed_equals_eager" 2>&1 | tail -3
timeout 300 python bench.py --batch 32 --steps 20 --warmup 4 --no-cpu-baseline > gpurun_out/r6b/base_b32_nopad.json 2> gpurun_out/r6b/base_b32_nopad.err < /dev/null
python -c "import json;d=json.load(open('gpurun_out/r6b/base_b32_nopad.json'));print('base B=32 nopad:', d['ms_per_step'], 'ms')"
unset ST5_HIP_LIB
timeout 300 python bench.py --batch 32 --steps 20 --warmup 4 --no-cpu-baseline > gpurun_out/r6b/base_b32_pad.json 2> gpurun_out/r6b/base_b32_pad.err < /dev/null
python -c "import json;d=json.load(open('gpurun_out/r6b/base_b32_pad.json'));print('base B=32 pad:', d['ms_per_step'], 'ms')"

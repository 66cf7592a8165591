#!/bin/bash
mkdir -p gpurun_out/r6b
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_bench_update_gpu.py tests/test_layerdrop_gpu.py -x -q -k "layernorm or layer_norm or ln or replayed_equals_eager or layerdrop" 2>&1 | tail -3
for rep in 1 2 3; do
for lib in new prev; do
  if [ $lib = prev ]; then export ST5_HIP_LIB=$PWD/speecht5_amd/libspeecht5_hip_prev.so; else unset ST5_HIP_LIB; fi
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r6b/base_lnoff_${lib}_$rep.json 2> gpurun_out/r6b/base_lnoff_${lib}_$rep.err < /dev/null
  python -c "import json;d=json.load(open('gpurun_out/r6b/base_lnoff_${lib}_$rep.json'));print('base LN row offsets, lib $lib rep $rep:', d['ms_per_step'], 'ms')"
done; done
unset ST5_HIP_LIB

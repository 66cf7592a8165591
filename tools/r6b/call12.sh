#!/bin/bash
mkdir -p gpurun_out/r6b
timeout 900 python -m pytest tests/test_flash_gpu.py tests/test_bench_update_gpu.py -x -q -k "not cfg4" 2>&1 | tail -3
python tools/bench_flash.py 2>&1 | tail -12
for rep in 1 2 3; do
for lib in new prev; do
  if [ $lib = prev ]; then export ST5_HIP_LIB=$PWD/speecht5_amd/libspeecht5_hip_prev.so; else unset ST5_HIP_LIB; fi
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r6b/base_qpt_${lib}_$rep.json 2> gpurun_out/r6b/base_qpt_${lib}_$rep.err < /dev/null
  python -c "import json;d=json.load(open('gpurun_out/r6b/base_qpt_${lib}_$rep.json'));print('base qp_table PE in LDS, lib $lib rep $rep:', d['ms_per_step'], 'ms')"
done; done
unset ST5_HIP_LIB

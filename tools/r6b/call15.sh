#!/bin/bash
mkdir -p gpurun_out/r6b
timeout 900 python -m pytest tests/test_bf16_path_gpu.py tests/test_bench_update_gpu.py -x -q -k "group or tn or replayed_equals_eager" 2>&1 | tail -3 > gpurun_out/r6b/xcd_tests_tail.txt; cat gpurun_out/r6b/xcd_tests_tail.txt
for rep in 1 2 3; do
for lib in new prev; do
  if [ $lib = prev ]; then export ST5_HIP_LIB=$PWD/speecht5_amd/libspeecht5_hip_prev.so; else unset ST5_HIP_LIB; fi
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r6b/base_xcd_${lib}_$rep.json 2> gpurun_out/r6b/base_xcd_${lib}_$rep.err < /dev/null
  python -c "import json;d=json.load(open('gpurun_out/r6b/base_xcd_${lib}_$rep.json'));print('base launch-global XCD order, lib $lib rep $rep:', d['ms_per_step'], 'ms')"
done; done
for lib in new prev; do
  if [ $lib = prev ]; then export ST5_HIP_LIB=$PWD/speecht5_amd/libspeecht5_hip_prev.so; else unset ST5_HIP_LIB; fi
  timeout 400 python bench.py --arch large --batch 32 --steps 15 --warmup 4 --no-cpu-baseline > gpurun_out/r6b/large_xcd_${lib}.json 2> gpurun_out/r6b/large_xcd_${lib}.err < /dev/null
  python -c "import json;d=json.load(open('gpurun_out/r6b/large_xcd_${lib}.json'));print('large bf16 B=32, lib $lib:', d['ms_per_step'], 'ms')"
done
unset ST5_HIP_LIB
timeout 500 bash tools/pmc_traffic.sh r6b_xcd < /dev/null | grep "tn8p_group"

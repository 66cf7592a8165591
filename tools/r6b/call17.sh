#!/bin/bash
mkdir -p gpurun_out/r6b
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_bf16_path_gpu.py tests/test_bench_update_gpu.py -x -q -k "tn or group or conv or replayed_equals_eager" 2>&1 | tail -3 > gpurun_out/r6b/skxcd_tests_tail.txt; cat gpurun_out/r6b/skxcd_tests_tail.txt
for rep in 1 2 3; do
for lib in new prev; do
  if [ $lib = prev ]; then export ST5_HIP_LIB=$PWD/speecht5_amd/libspeecht5_hip_prev.so; else unset ST5_HIP_LIB; fi
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r6b/base_skxcd_${lib}_$rep.json 2> gpurun_out/r6b/base_skxcd_${lib}_$rep.err < /dev/null
  python -c "import json;d=json.load(open('gpurun_out/r6b/base_skxcd_${lib}_$rep.json'));print('base split-K XCD placement, lib $lib rep $rep:', d['ms_per_step'], 'ms')"
done; done
unset ST5_HIP_LIB
timeout 500 bash tools/pmc_traffic.sh r6b_skxcd < /dev/null | grep "tn_glds"
python - <<'PY'
import json
p=json.load(open('gpurun_out/pmc/r6b_skxcd_pmc_traffic.json'))
for k,v in p['kernels'].items():
    if "tn_glds" in k: print(k[:60], v)
PY

#!/bin/bash
mkdir -p gpurun_out/r6b
timeout 900 python -m pytest tests/test_flash_gpu.py tests/test_bf16_path_gpu.py tests/test_bench_update_gpu.py -x -q -k "flash or attention or replayed_equals_eager" 2>&1 | tail -3 > gpurun_out/r6b/attn_xcd_tests_tail.txt; cat gpurun_out/r6b/attn_xcd_tests_tail.txt
for rep in 1 2 3; do
for lib in new prev; do
  if [ $lib = prev ]; then export ST5_HIP_LIB=$PWD/speecht5_amd/libspeecht5_hip_prev.so; else unset ST5_HIP_LIB; fi
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r6b/base_axcd_${lib}_$rep.json 2> gpurun_out/r6b/base_axcd_${lib}_$rep.err < /dev/null
  python -c "import json;d=json.load(open('gpurun_out/r6b/base_axcd_${lib}_$rep.json'));print('base attention XCD mapping, lib $lib rep $rep:', d['ms_per_step'], 'ms')"
done; done
unset ST5_HIP_LIB
timeout 500 bash tools/pmc_traffic.sh r6b_axcd < /dev/null | grep "fa2"
python - <<'PY'
import json
p=json.load(open('gpurun_out/pmc/r6b_axcd_pmc_traffic.json'))
for k,v in p['kernels'].items():
    if 'fa2' in k: print(k[:60], v)
PY

#!/bin/bash
# whole-update A/B of the cache policy of the 128^2 NT kernel's operand loads (same box, alternating)
mkdir -p gpurun_out/r6
for rep in 1 2; do
for v in base aux2_2 aux2_0 aux0_2 aux1_1; do
  lib=""; [ $v != base ] && lib="$PWD/speecht5_amd/libspeecht5_hip_$v.so"
  ST5_HIP_LIB=$lib timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r6/ab_$v_$rep.json 2> gpurun_out/r6/ab_$v_$rep.err
  python -c "import json;d=json.load(open('gpurun_out/r6/ab_$v_$rep.json'));print('$v rep $rep:', d['ms_per_step'], 'ms; nt frac', d['roofline']['frac'])" || tail -3 gpurun_out/r6/ab_$v_$rep.err
done; done

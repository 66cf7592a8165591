#!/bin/bash
mkdir -p gpurun_out/r6
for rep in 1 2; do
for lim in 0 130 200; do
  ST5_M64_MAX_TILES=$lim timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r6/ab2_m64_${lim}_$rep.json 2> gpurun_out/r6/ab2_m64_${lim}_$rep.err
  python -c "import json;d=json.load(open('gpurun_out/r6/ab2_m64_${lim}_$rep.json'));print('m64 max tiles $lim rep $rep:', d['ms_per_step'], 'ms')"
done; done

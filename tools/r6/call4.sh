#!/bin/bash
# whole-update A/B of the half-height NT tiles: same box, alternating
mkdir -p gpurun_out/r6
for rep in 1 2; do
for lim in 0 400 640 1200; do
  ST5_M64_MAX_TILES=$lim timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r6/ab_m64_${lim}_$rep.json 2> gpurun_out/r6/ab_m64_${lim}_$rep.err
  python -c "import json;d=json.load(open('gpurun_out/r6/ab_m64_${lim}_$rep.json'));print('m64 max tiles $lim rep $rep:', d['ms_per_step'], 'ms; roofline.frac', d['roofline']['frac'], 'nt ms', d['roofline']['all_variants']['bf16_NT']['ms_per_step'])"
done; done

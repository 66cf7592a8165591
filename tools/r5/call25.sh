#!/bin/bash
# Large (cfg 5) at B = 32 per GPU, bf16 / fp8 alternating on one box, at the end-of-round kernels
mkdir -p gpurun_out/r5x
for tag in bf16 fp8 bf16_again fp8_again; do
  extra=""; case $tag in fp8*) extra="--dtype fp8";; esac
  timeout 300 python bench.py --arch large --batch 32 --steps 10 --warmup 5 --no-cpu-baseline $extra > gpurun_out/r5x/r5_large_b32_$tag.json 2> gpurun_out/r5x/r5_large_b32_$tag.err
  python -c "import json;d=json.load(open('gpurun_out/r5x/r5_large_b32_$tag.json'));print('$tag', d['ms_per_step'], d['value'], d['roofline']['frac'])"
done

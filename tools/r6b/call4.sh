#!/bin/bash
mkdir -p gpurun_out/r6b
timeout 900 python -m pytest tests/test_bf16_path_gpu.py -x -q -k "group or tn" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_fp8_gpu.py -x -q -k "not oracle" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_bench_update_gpu.py -x -q -k "replayed_equals_eager" 2>&1 | tail -4
for rep in 1 2; do
for tg in 0 1; do
  ST5_TN_GROUP_TILE=$tg timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r6b/base_tg${tg}_$rep.json 2> gpurun_out/r6b/base_tg${tg}_$rep.err < /dev/null
  python -c "import json;d=json.load(open('gpurun_out/r6b/base_tg${tg}_$rep.json'));print('base tn-group-tile $tg rep $rep:', d['ms_per_step'], 'ms')"
done; done
for rep in 1 2; do
for arm in bf16_tg0 bf16_tg1 fp8_h16_f1 fp8_h16_f0 fp8_h0_f1; do
  case $arm in
    bf16_tg0) dt=bf16; env="ST5_TN_GROUP_TILE=0";;
    bf16_tg1) dt=bf16; env="ST5_TN_GROUP_TILE=1";;
    fp8_h16_f1) dt=fp8; env="ST5_MX8_HEAVY_NK=16 ST5_FP8_FUSE_QUANT=1";;
    fp8_h16_f0) dt=fp8; env="ST5_MX8_HEAVY_NK=16 ST5_FP8_FUSE_QUANT=0";;
    fp8_h0_f1) dt=fp8; env="ST5_MX8_HEAVY_NK=0 ST5_FP8_FUSE_QUANT=1";;
  esac
  env $env timeout 400 python bench.py --arch large --batch 32 --dtype $dt --steps 15 --warmup 4 --no-cpu-baseline > gpurun_out/r6b/large_${arm}_$rep.json 2> gpurun_out/r6b/large_${arm}_$rep.err < /dev/null
  python -c "import json;d=json.load(open('gpurun_out/r6b/large_${arm}_$rep.json'));print('large $arm rep $rep:', d['ms_per_step'], 'ms')"
done; done

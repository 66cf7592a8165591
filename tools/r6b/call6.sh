#!/bin/bash
R=$PWD; mkdir -p $R/gpurun_out/r6b
timeout 900 python -m pytest tests/test_fp8_gpu.py -x -q 2>&1 | tail -4
for rep in 1 2; do
for f in 0 7 6; do
  ST5_FP8_FUSE_QUANT=$f timeout 400 python bench.py --arch large --batch 32 --dtype fp8 --steps 15 --warmup 4 --no-cpu-baseline > gpurun_out/r6b/large_fp8_fast_fuse${f}_$rep.json 2> gpurun_out/r6b/large_fp8_fast_fuse${f}_$rep.err < /dev/null
  python -c "import json;d=json.load(open('gpurun_out/r6b/large_fp8_fast_fuse${f}_$rep.json'));print('fast quantiser, fuse mask $f rep $rep:', d['ms_per_step'], 'ms')"
done
timeout 400 python bench.py --arch large --batch 32 --dtype bf16 --steps 15 --warmup 4 --no-cpu-baseline > gpurun_out/r6b/large_bf16_fast_$rep.json 2> gpurun_out/r6b/large_bf16_fast_$rep.err < /dev/null
python -c "import json;d=json.load(open('gpurun_out/r6b/large_bf16_fast_$rep.json'));print('bf16 rep $rep:', d['ms_per_step'], 'ms')"
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_f7
ST5_FP8_FUSE_QUANT=7 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_f7 -- python $R/bench.py --arch large --batch 32 --dtype fp8 --steps 6 --warmup 3 --no-cpu-baseline > /dev/null 2>&1 < /dev/null
cp $(ls /tmp/kt_f7/*/*kernel_stats.csv | head -1) $R/gpurun_out/r6b/large_fp8_fast_fuse7_kernel_stats.csv

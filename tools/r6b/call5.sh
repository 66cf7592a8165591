#!/bin/bash
# fp8 mode, Large B = 32: kernel-trace summaries by fusion mask (0 none, 1 LayerNorm, 2 GELU epilogue, 4 GELU-derivative epilogue, 7 all)
R=$PWD; mkdir -p $R/gpurun_out/r6b
cd /tmp && export TMPDIR=/tmp
for f in 0 7; do
  rm -rf /tmp/kt_f$f
  ST5_FP8_FUSE_QUANT=$f timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_f$f -- python $R/bench.py --arch large --batch 32 --dtype fp8 --steps 6 --warmup 3 --no-cpu-baseline > /dev/null 2>&1 < /dev/null
  cp $(ls /tmp/kt_f$f/*/*kernel_stats.csv | head -1) $R/gpurun_out/r6b/large_fp8_fuse${f}_kernel_stats.csv
done
cd $R
for rep in 1 2; do
for f in 0 1 2 4 7; do
  ST5_FP8_FUSE_QUANT=$f timeout 400 python bench.py --arch large --batch 32 --dtype fp8 --steps 15 --warmup 4 --no-cpu-baseline > gpurun_out/r6b/large_fp8_fuse${f}_$rep.json 2> gpurun_out/r6b/large_fp8_fuse${f}_$rep.err < /dev/null
  python -c "import json;d=json.load(open('gpurun_out/r6b/large_fp8_fuse${f}_$rep.json'));print('fuse mask $f rep $rep:', d['ms_per_step'], 'ms')"
done; done

#!/bin/bash
mkdir -p gpurun_out/r6b
timeout 900 python -m pytest tests/test_fp8_gpu.py -x -q 2>&1 | tail -15
for rep in 1 2; do
for mode in fp8 fp8nofuse bf16; do
  dt=$mode; fuse=1; if [ $mode = fp8nofuse ]; then dt=fp8; fuse=0; fi
  ST5_FP8_FUSE_QUANT=$fuse timeout 400 python bench.py --arch large --batch 32 --dtype $dt --steps 15 --warmup 4 --no-cpu-baseline > gpurun_out/r6b/large_b32_${mode}_q_$rep.json 2> gpurun_out/r6b/large_b32_${mode}_q_$rep.err < /dev/null
  python -c "import json;d=json.load(open('gpurun_out/r6b/large_b32_${mode}_q_$rep.json'));print('$mode rep $rep:', d['ms_per_step'], 'ms')"
done; done
R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt_l8
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_l8 -- python $R/bench.py --arch large --batch 32 --dtype fp8 --steps 6 --warmup 3 --no-cpu-baseline > /dev/null 2>&1 < /dev/null
cp $(ls /tmp/kt_l8/*/*kernel_stats.csv | head -1) $R/gpurun_out/r6b/large_b32_fp8_fused_kernel_stats.csv

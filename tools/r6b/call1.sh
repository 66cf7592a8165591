#!/bin/bash
# (1) which hipBLASLt kernels run the transformer's NT shapes (macro tile, stream-K?) -- kernel trace of tools/gemm_cases.py nt
# (2) Large B = 32: kernel-trace summaries of the bf16 and the fp8 update (where does the fp8 mode's time go?)
R=$PWD; mkdir -p $R/gpurun_out/r6b
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_gc
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_gc -- python $R/tools/gemm_cases.py nt > $R/gpurun_out/r6b/gemm_cases_nt.txt 2>&1 < /dev/null
f=$(ls /tmp/kt_gc/*/*kernel_stats.csv | head -1); cp "$f" $R/gpurun_out/r6b/gemm_cases_kernel_stats.csv
grep -i "Cijk\|MT[0-9]" "$f" | cut -c1-700 > $R/gpurun_out/r6b/hipblaslt_kernels.txt
for dt in bf16 fp8; do
  rm -rf /tmp/kt_l$dt
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_l$dt -- python $R/bench.py --arch large --batch 32 --dtype $dt --steps 6 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r6b/large_b32_$dt.json 2> $R/gpurun_out/r6b/large_b32_$dt.err < /dev/null
  f=$(ls /tmp/kt_l$dt/*/*kernel_stats.csv | head -1); cp "$f" $R/gpurun_out/r6b/large_b32_${dt}_kernel_stats.csv
done
cd $R
head -c 400 gpurun_out/r6b/large_b32_bf16.json; echo; head -c 400 gpurun_out/r6b/large_b32_fp8.json; echo
cat gpurun_out/r6b/hipblaslt_kernels.txt | cut -c1-400

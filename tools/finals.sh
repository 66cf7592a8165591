#!/bin/bash
# Round-end measurements in ONE GPU call (every step under its own timeout; nothing here reads stdin):
#   gpurun_out/${TAG}_bench_kernel_stats.csv        rocprofv3 --kernel-trace --stats of a 13-step bench run
#   gpurun_out/pmc/${TAG}_pmc_traffic.json          tools/pmc_traffic.sh (two --pmc passes)
#   gpurun_out/${TAG}_bench.json / .err             python bench.py (driver defaults) -- LAST of the three, with the two summaries above
#                                                   already copied into the box's profiles/, so that its `roofline.traffic` and
#                                                   `from_kernel_trace` come from THIS run's kernels (bench.py checks the source hash)
#   gpurun_out/${TAG}_cfg3.json                     python bench.py --config 3
#   gpurun_out/${TAG}_bench_b32.json                python bench.py --batch 32 (cfg 4's per-GPU shape)
# Copy what should be judged into profiles/.
TAG=${1:-r4}
R=$PWD
mkdir -p $R/gpurun_out
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_$TAG && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$TAG -- python $R/bench.py --steps 13 --warmup 5 --no-cpu-baseline > /tmp/kt_$TAG.log 2>&1 < /dev/null; echo "rocprof rc=$?"
  f=$(ls /tmp/kt_$TAG/*/*kernel_stats.csv 2>/dev/null | head -1); if [ -n "$f" ]; then cp "$f" $R/gpurun_out/${TAG}_bench_kernel_stats.csv; python -c "import sys;sys.path.insert(0,'$R');import hashlib,json;json.dump({'gemm_hip_sha1':__import__('bench').kernel_source_hash(),'command':'rocprofv3 --kernel-trace --stats -- python bench.py --steps 13 --warmup 5 --no-cpu-baseline'},open('$R/gpurun_out/${TAG}_bench_kernel_stats.meta.json','w'))"; else echo "no kernel_stats.csv"; ls -R /tmp/kt_$TAG | head -20; fi )
timeout 500 bash tools/pmc_traffic.sh $TAG < /dev/null | tail -4; echo "pmc rc=$?"
cp gpurun_out/${TAG}_bench_kernel_stats.csv gpurun_out/${TAG}_bench_kernel_stats.meta.json profiles/ 2>/dev/null
cp gpurun_out/pmc/${TAG}_pmc_traffic.json profiles/ 2>/dev/null
timeout 400 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err < /dev/null; echo "bench rc=$?"
timeout 300 python bench.py --config 3 --steps 20 --warmup 5 > gpurun_out/${TAG}_cfg3.json 2> gpurun_out/${TAG}_cfg3.err < /dev/null; echo "cfg3 rc=$?"
timeout 300 python bench.py --batch 32 --no-cpu-baseline > gpurun_out/${TAG}_bench_b32.json 2> gpurun_out/${TAG}_bench_b32.err < /dev/null; echo "b32 rc=$?"
head -c 600 gpurun_out/${TAG}_bench.json; echo

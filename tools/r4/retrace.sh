#!/bin/bash
# kernel trace + PMC traffic of the default bench for the CURRENT csrc/gemm.hip (the two summaries bench.py reads from profiles/)
TAG=${1:-r4}
R=$PWD
mkdir -p $R/gpurun_out
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_$TAG && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$TAG -- python $R/bench.py --steps 13 --warmup 5 --no-cpu-baseline > /tmp/kt_$TAG.log 2>&1 < /dev/null; echo "rocprof rc=$?"
  f=$(ls /tmp/kt_$TAG/*/*kernel_stats.csv 2>/dev/null | head -1); if [ -n "$f" ]; then cp "$f" $R/gpurun_out/${TAG}_bench_kernel_stats.csv; python -c "import hashlib,json;json.dump({'gemm_hip_sha1':hashlib.sha1(open('$R/speecht5_amd/csrc/gemm.hip','rb').read()).hexdigest()[:12],'command':'rocprofv3 --kernel-trace --stats -- python bench.py --steps 13 --warmup 5 --no-cpu-baseline'},open('$R/gpurun_out/${TAG}_bench_kernel_stats.meta.json','w'))"; else echo "no kernel_stats.csv"; fi )
timeout 500 bash tools/pmc_traffic.sh $TAG < /dev/null | tail -3; echo "pmc rc=$?"

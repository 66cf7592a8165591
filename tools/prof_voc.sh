#!/bin/bash
# rocprofv3 kernel stats of the vocoder alone (tools/bench_cfg3.py, CFG3_VOCODER_ONLY=1) -> gpurun_out/voc_kernel_stats.csv
R=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/voc_kt
CFG3_VOCODER_ONLY=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/voc_kt -- python $R/tools/bench_cfg3.py > /tmp/voc_kt.log 2>&1 < /dev/null
f=$(ls /tmp/voc_kt/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $R/gpurun_out/voc_kernel_stats.csv; head -12 "$f" | cut -c1-150; else echo "no stats file"; tail -5 /tmp/voc_kt.log; fi

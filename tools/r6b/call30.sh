#!/bin/bash
# conv layer 0 forward: GELU through the LDS chord table against the polynomial -- parity, then device time A/B on one box
mkdir -p gpurun_out/r6b
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k conv0 2>&1 | grep -E "passed|failed|Error|assert" | tail -6
for f in 0 1 0 1; do
  ST5_T=$f timeout 300 python - <<PY
import os, torch, bench
from speecht5_amd import hip
hip.lib().st5_conv0_set_gelu_table(int(os.environ["ST5_T"]))
r = bench.conv0_device_time(torch.device("cuda:0"), 8, reps=200)
print("gelu table", os.environ["ST5_T"], r)
PY
done 2>&1 | grep "gelu table" | tee gpurun_out/r6b/conv0_gelu_table_ab.txt
cd /tmp && export TMPDIR=/tmp
for f in 0 1; do
  ST5_T=$f rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$f -o p -- python - <<PY > /dev/null 2>&1
import os, sys
sys.path.insert(0, "/root/repo"); os.chdir("/root/repo")
import torch, bench
from speecht5_amd import hip
hip.lib().st5_conv0_set_gelu_table(int(os.environ["ST5_T"]))
bench.conv0_device_time(torch.device("cuda:0"), 8, reps=100)
PY
  f2=$(find /tmp/prof_$f -name "*kernel_stats.csv" | head -1); [ -z "$f2" ] && { find /tmp/prof_$f | head; continue; }
  cp "$f2" /root/repo/gpurun_out/r6b/conv0_table${f}_kernel_stats.csv
done

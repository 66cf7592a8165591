#!/bin/bash
# conv layer 0: statistics + weight fragments folded into the apply kernel's prologue -- parity, then the device time A/B on one box
mkdir -p gpurun_out/r6b
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k conv0 2>&1 | tail -4
timeout 900 python -m pytest tests/test_bench_update_gpu.py tests/test_graph_gpu.py -m gpu -q 2>&1 | tail -2
for f in 0 1 0 1; do
  ST5_CONV0_FOLD=$f timeout 300 python - <<PY
import os, torch, bench
from speecht5_amd import hip
hip.lib().st5_conv0_set_fold(int(os.environ["ST5_CONV0_FOLD"]))
r = bench.conv0_device_time(torch.device("cuda:0"), 8, reps=200)
print("fold", os.environ["ST5_CONV0_FOLD"], r)
PY
done 2>&1 | grep fold | tee gpurun_out/r6b/conv0_fold_ab.txt
cd /tmp && export TMPDIR=/tmp
for f in 0 1; do
  ST5_CONV0_FOLD=$f rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$f -o p -- python - <<PY > /dev/null 2>&1
import os, sys
sys.path.insert(0, "/root/repo"); os.chdir("/root/repo")
import torch, bench
from speecht5_amd import hip
hip.lib().st5_conv0_set_fold(int(os.environ["ST5_CONV0_FOLD"]))
bench.conv0_device_time(torch.device("cuda:0"), 8, reps=100)
PY
  f2=$(find /tmp/prof_$f -name "*kernel_stats.csv" | head -1); [ -z "$f2" ] && { find /tmp/prof_$f | head; continue; }
  echo "== fold $f"; head -8 "$f2" | cut -d, -f1-8 | sed 's/(.*)//' 
  cp "$f2" $GRAFT_REPO_ROOT/gpurun_out/r6b/conv0_fold${f}_kernel_stats.csv
done

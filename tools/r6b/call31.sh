#!/bin/bash
# timing experiment: the forward apply kernel with lane-consecutive 16-byte stores (wrong placement, same bytes) against the real store pattern
mkdir -p gpurun_out/r6b
cd /tmp && export TMPDIR=/tmp
for f in real fake; do
  [ $f = fake ] && export ST5_HIP_LIB=/root/repo/tools/r6b/_libfake.so
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$f -o p -- python - <<PY > /tmp/log_$f 2>&1
import os, sys
sys.path.insert(0, "/root/repo"); os.chdir("/root/repo")
import torch, bench
bench.conv0_device_time(torch.device("cuda:0"), 8, reps=100)
PY
  f2=$(find /tmp/prof_$f -name "*kernel_stats.csv" | head -1); [ -z "$f2" ] && { tail -5 /tmp/log_$f; continue; }
  echo "== $f"; grep -E "apply_mfma" "$f2" | sed 's/(.*)//' | cut -d, -f1-4 | cut -c1-200
  cp "$f2" /root/repo/gpurun_out/r6b/conv0_store_${f}_kernel_stats.csv
done

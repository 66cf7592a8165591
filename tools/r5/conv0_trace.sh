#!/bin/bash
# kernel trace of conv layer 0's forward / backward alone (bench.conv0_device_time: 23 back-to-back calls each at the benched shape)
O=gpurun_out/r5k; mkdir -p $O; R=$PWD
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt0 && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt0 -- python -c "
import sys; sys.path.insert(0, '$R')
import torch, bench
print(bench.conv0_device_time(torch.device('cuda:0'), 8))
" > /tmp/kt0.log 2>&1; tail -2 /tmp/kt0.log | cut -c1-200
  f=$(ls /tmp/kt0/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $R/$O/conv0_kernel_stats.csv )
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r5k/conv0_kernel_stats.csv')):
    if 'conv0' in r['Name']: print(f"{float(r['AverageNs'])/1e3:8.1f} us x{int(r['Calls'])}  {r['Name'][:70]}")
PY

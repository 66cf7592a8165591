#!/bin/bash
# GPU call 9: kernel trace of the bench (conv0 kernels' own durations inside replayed updates) + the full-size test again
O=gpurun_out/r5i; mkdir -p $O
timeout 600 python -m pytest tests/test_fullsize_gpu.py -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
R=$PWD
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --steps 13 --warmup 5 --no-cpu-baseline > /tmp/kt.log 2>&1 < /dev/null; echo "rocprof rc=$?"
  f=$(ls /tmp/kt/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $R/$O/kernel_stats.csv )
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r5i/kernel_stats.csv')))
upd=sum(int(r['Calls']) for r in rows if 'adam_kernel' in r['Name'])
print('updates',upd,'total kernel ms/upd',sum(float(r['TotalDurationNs']) for r in rows)/upd/1e6)
for r in rows:
    if 'conv0' in r['Name']: print(f"{float(r['AverageNs'])/1e3:8.1f} us x{int(r['Calls'])/upd:.1f}  {r['Name'][:80]}")
PY

#!/bin/bash
O=gpurun_out/r5r; mkdir -p $O
timeout 900 python -m pytest tests/test_flash_gpu.py tests/test_bf16_path_gpu.py tests/test_bench_update_gpu.py::test_benched_update_replayed_equals_eager_and_reproduces -q -m gpu > $O/tests.log 2>&1; tail -2 $O/tests.log
R=$PWD
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --steps 13 --warmup 5 --no-cpu-baseline > /tmp/kt.log 2>&1 < /dev/null
  f=$(ls /tmp/kt/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $R/$O/kernel_stats.csv; tail -1 /tmp/kt.log | cut -c1-300 )
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r5r/kernel_stats.csv')))
upd=sum(int(r['Calls']) for r in rows if 'adam_kernel' in r['Name'])
print('updates',upd,'kernel ms/upd',sum(float(r['TotalDurationNs']) for r in rows)/upd/1e6)
for r in rows:
    if 'fa2::' in r['Name']: print(f"{float(r['AverageNs'])/1e3:8.1f} us x{int(r['Calls'])/upd:.1f}  {r['Name'][:70]}")
PY

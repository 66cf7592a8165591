#!/bin/bash
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ln_a /tmp/ln_kt
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ln_kt -- python $R/tools/r5/ln_time.py > /dev/null 2>&1 < /dev/null
timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_WAVES --output-format csv -d /tmp/ln_a -- python $R/tools/r5/ln_time.py > /dev/null 2>&1 < /dev/null
python - <<'PY'
import csv, glob, collections
for row in csv.DictReader(open(glob.glob('/tmp/ln_kt/**/*kernel_stats.csv', recursive=True)[0])):
    if 'ln_' in row['Name']: print(f"{float(row['AverageNs'])/1e3:8.1f} us x{row['Calls']}  {row['Name'][:90]}")
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(glob.glob('/tmp/ln_a/**/*counter_collection.csv', recursive=True)[0])):
    k = row['Kernel_Name'][:60]
    if 'ln_' not in k: continue
    acc[k][row['Counter_Name']] += float(row['Counter_Value'])
    if row['Counter_Name'] == 'SQ_WAVES': n[k] += 1
for k, v in acc.items():
    print(k, n[k], {c: round(x / max(n[k], 1)) for c, x in v.items()})
PY

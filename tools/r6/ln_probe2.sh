#!/bin/bash
# device time of ln_bwd_vec_kernel by variant (kernel trace: the host-side cost of a ctypes call hides it from event timing)
R=$PWD; cd /tmp && export TMPDIR=/tmp
for v in "256 8192 1 0" "256 8192 0 0" "256 8192 1 1" "512 8192 1 0" "512 8192 0 0" "512 8192 1 1" "256 3992 1 1" "512 3992 1 1" "1024 8192 1 1"; do
  set -- $v
  rm -rf /tmp/lnp
  LN_ONE="$1 $2 $3 $4" LN_DEFER=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lnp -- python $R/tools/r6/ln_probe.py > /tmp/lnp.log 2>&1
  f=$(ls /tmp/lnp/*/*kernel_stats.csv | head -1)
  python - <<PY
import csv
for r in csv.DictReader(open("$f")):
    if "ln_bwd_vec" in r["Name"]:
        print("cap $1 rows $2 partials $3 dropped-copy $4:", r["Calls"], "calls, avg", round(float(r["AverageNs"])/1e3, 2), "us")
PY
done

#!/bin/bash
# kernel-trace timing of the fused attention kernels (tools/flash_pmc.py: B 16, H 12, T 512) for the product library [+ variants]
R=$PWD; O=$R/gpurun_out/${1:-r4c}; mkdir -p $O; shift
cd /tmp && export TMPDIR=/tmp
for v in base "$@"; do
  lib=$R/speecht5_amd/libspeecht5_hip.so; [ $v != base ] && lib=$R/speecht5_amd/libspeecht5_hip_$v.so
  for rel in 1 0; do
    rm -rf /tmp/fl_$v$rel
    ST5_HIP_LIB=$lib REL=$rel ITERS=10 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fl_$v$rel -- python $R/tools/flash_pmc.py > /dev/null 2>&1 < /dev/null
    f=$(find /tmp/fl_$v$rel -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && python - "$f" "$v rel=$rel" <<'PY' | tee -a $O/fa_time.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "fa2::" in r["Name"]:
        print(f"{sys.argv[2]:14s} {r['Name'][:58]:58s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e3:7.1f} us")
PY
  done
done

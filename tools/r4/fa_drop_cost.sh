#!/bin/bash
# what the dropout draw costs the attention kernels: kernel-trace averages with p = 0.1 and p = 0 (the DROP = false instantiations)
R=$PWD; O=$R/gpurun_out/${1:-fad}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for p in 0.1 0; do for rel in 1 0; do
  rm -rf /tmp/fd_$p$rel
  PDROP=$p REL=$rel ITERS=10 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fd_$p$rel -- python $R/tools/flash_pmc.py > /dev/null 2>&1 < /dev/null
  f=$(find /tmp/fd_$p$rel -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python - "$f" "p=$p rel=$rel" <<'PY' | tee -a $O/fa_drop.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "fa2::" in r["Name"] and "qp_table" not in r["Name"]:
        print(f"{sys.argv[2]:14s} {r['Name'][:58]:58s} avg {float(r['AverageNs'])/1e3:7.1f} us")
PY
done; done

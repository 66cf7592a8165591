#!/bin/bash
# round-4 GPU call B: measurements only -- attention ablations (kernel trace per library variant), GEMM class table, hazard reproducer
set -u
R=$PWD; O=$R/gpurun_out/r4b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
$R/tools/hazard/pk_hazard.bin 16 > $O/pk_hazard.json 2> $O/pk_hazard.err; cat $O/pk_hazard.json | cut -c1-1500
for v in base fa1 fa2 fa3 fa4; do
  lib=$R/speecht5_amd/libspeecht5_hip.so; [ $v != base ] && lib=$R/speecht5_amd/libspeecht5_hip_$v.so
  for rel in 1 0; do
    rm -rf /tmp/fl_$v$rel
    ST5_HIP_LIB=$lib REL=$rel ITERS=10 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fl_$v$rel -- python $R/tools/flash_pmc.py > /dev/null 2>&1 < /dev/null
    f=$(find /tmp/fl_$v$rel -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && grep "fa2::" $f | awk -F'","' -v tag="$v rel=$rel" '{gsub(/"/,"",$1); printf "%s | %-60s calls %s avg %.1f us\n", tag, substr($1,1,60), $2, $4/1000}' | tee -a $O/fa_abl.txt
  done
done
cd $R
timeout 300 python tools/gemm_cases.py > $O/gemm_cases.txt 2>&1; grep -E "^(NT|TN)" $O/gemm_cases.txt

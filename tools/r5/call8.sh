#!/bin/bash
# GPU call 8 of round 5: conv layer 0 forward + backward on the matrix cores, folded-scale GELU, two-launch statistics.
O=gpurun_out/r5h; mkdir -p $O
echo tests skipped
run() { echo -n "$1: "; env $2 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $3 2>$O/err.log | python tools/r5/conv0_line.py; }
{
run "matrix-core conv0" "A=1" ""
run "VALU conv0" "ST5_CONV0_MFMA=0" ""
run "matrix-core conv0" "A=1" ""
run "VALU conv0" "ST5_CONV0_MFMA=0" ""
} > $O/conv0.log 2>&1
cat $O/conv0.log

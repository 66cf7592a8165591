#!/bin/bash
# failure rate of the replayed side-by-side update with clipping ON (fresh processes)
O=gpurun_out/${1:-r4h}; mkdir -p $O; N=${2:-6}; REPS=${3:-24}
timeout 120 python tools/one_run.py graph in_turn $N 0.05 2>/dev/null | grep RESULT | tee -a $O/sbs_rate.txt
for i in $(seq $REPS); do timeout 120 python tools/one_run.py graph side_by_side $N 0.05 2>/dev/null | grep RESULT | tee -a $O/sbs_rate.txt; done
sort $O/sbs_rate.txt | awk '{print $2, $3, $7}' | uniq -c

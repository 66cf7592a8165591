#!/bin/bash
# determinism matrix of the side-by-side update at full size, fresh processes (hash of parameters + moments after N updates)
O=gpurun_out/${1:-r4e}; mkdir -p $O; N=${2:-4}; REPS=${3:-5}
for mode in in_turn in_turn_2buf; do timeout 120 python tools/one_run.py eager $mode $N 0.05 2>/dev/null | grep RESULT | tee -a $O/sbs_matrix.txt; done
for i in $(seq $REPS); do timeout 120 python tools/one_run.py eager side_by_side $N 0.05 2>/dev/null | grep RESULT | tee -a $O/sbs_matrix.txt; done
for i in $(seq $REPS); do timeout 120 python tools/one_run.py graph side_by_side $N 0.05 2>/dev/null | grep RESULT | tee -a $O/sbs_matrix.txt; done
timeout 120 python tools/one_run.py graph in_turn $N 0.05 2>/dev/null | grep RESULT | tee -a $O/sbs_matrix.txt

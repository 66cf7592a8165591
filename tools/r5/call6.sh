#!/bin/bash
# GPU call 6 of round 5: the GPU collater test; does a stream priority / the order of the micro-batches shorten the side-by-side update?
O=gpurun_out/r5f; mkdir -p $O
echo skip tests
run() { echo -n "$1: "; env $2 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $3 2>$O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], d['unit'])"; }
{
run "default" "A=1" ""
run "text stream high priority" "ST5_SBS_PRIORITY=-1" ""
run "text first (owns the update's stream)" "ST5_TEXT_FIRST=1" ""
run "text first + speech stream high priority" "ST5_TEXT_FIRST=1 ST5_SBS_PRIORITY=-1" ""
run "default again" "A=1" ""
run "text stream high priority again" "ST5_SBS_PRIORITY=-1" ""
} > $O/prio.log 2>&1
cat $O/prio.log

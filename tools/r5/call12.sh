#!/bin/bash
# GPU call 12: a THIRD concurrent chain -- the weight-gradient GEMMs of the micro-batch that owns the update's stream on a helper stream
# forked from the capture's first parent.  Same bits?  Faster?
O=gpurun_out/r5o; mkdir -p $O
H="python tools/r5/replay_hunt.py"; G='HUNT|DIFF|Error|error|assert|differ|identical'
{
timeout 200 $H run $O/ref.json --n 120 2>&1 | grep -E "$G"
ST5_SBS_OWNER=1 timeout 200 $H run $O/own1w.json --n 120 --mode side_by_side --wgrad 1 2>&1 | grep -E "$G"; $H diff $O/ref.json $O/own1w.json
ST5_SBS_OWNER=0 timeout 200 $H run $O/own0w.json --n 120 --mode side_by_side --wgrad 1 2>&1 | grep -E "$G"; $H diff $O/ref.json $O/own0w.json
} > $O/A.log 2>&1
cat $O/A.log
run() { echo -n "$1: "; env $2 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $3 2>$O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'])"; }
{
run "default" "A=1" ""
run "text owns the update's stream" "ST5_SBS_OWNER=1" ""
run "text owns it + its weight gradients on a helper stream" "ST5_SBS_OWNER=1 ST5_WGRAD_STREAM=1" ""
run "speech owns it + its weight gradients on a helper stream" "ST5_SBS_OWNER=0 ST5_WGRAD_STREAM=1" ""
run "default" "A=1" ""
run "text owns it + its weight gradients on a helper stream" "ST5_SBS_OWNER=1 ST5_WGRAD_STREAM=1" ""
run "B=32: default" "A=1" "--batch 32 --steps 10"
run "B=32: speech owns it + helper stream" "ST5_SBS_OWNER=0 ST5_WGRAD_STREAM=1" "--batch 32 --steps 10"
} > $O/B.log 2>&1
cat $O/B.log; tail -2 $O/err.log | cut -c1-200

#!/bin/bash
# GPU call 10: a weight-gradient stream PER micro-batch stream (4 streams): same bits? faster?
O=gpurun_out/r5j; mkdir -p $O
H="python tools/r5/replay_hunt.py"; G='HUNT|DIFF|Error|error|assert|differ|identical'
{
timeout 200 $H run $O/ref.json --n 100 2>&1 | grep -E "$G"
timeout 200 $H run $O/sbsw.json --n 100 --mode side_by_side --wgrad 1 2>&1 | grep -E "$G"; $H diff $O/ref.json $O/sbsw.json
} > $O/A.log 2>&1
cat $O/A.log
run() { echo -n "$1: "; env $2 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $3 2>$O/err.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); h=d['roofline']['hbm_bound_kernels']; print(d['ms_per_step'], 'ms', d['value'], '| conv0 device fwd', h['conv0_gn_gelu_fwd'].get('device_ms'), h['conv0_gn_gelu_fwd'].get('device_frac_of_8TBps'), 'bwd', h['conv0_gn_gelu_bwd'].get('device_ms'), h['conv0_gn_gelu_bwd'].get('device_frac_of_8TBps'))"; }
{
run "default" "A=1" ""
run "wgrad stream per micro-batch" "ST5_WGRAD_STREAM=1" ""
run "default" "A=1" ""
run "wgrad stream per micro-batch" "ST5_WGRAD_STREAM=1" ""
run "wgrad streams, conv too off" "ST5_WGRAD_STREAM=1 ST5_WGRAD_CONV=0" ""
} > $O/B.log 2>&1
cat $O/B.log; tail -3 $O/err.log

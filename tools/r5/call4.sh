#!/bin/bash
# GPU call 4 of round 5: the tests this round added / changed for the side-by-side default; knob A/B of the GEMM tile policies in the
# side-by-side mode (the other stream now fills tile-quantisation tails: do the 256^2 kernels pay on more shapes?); a kernel trace of
# the side-by-side replay; the per-GPU shape of cfg 4 (B = 32).
O=gpurun_out/r5d; mkdir -p $O
timeout 1200 python -m pytest tests/test_replay_long_gpu.py tests/test_bench_update_gpu.py tests/test_graph_gpu.py tests/test_two_rank_gpu.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
run() { echo -n "$1: "; env $2 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], d['unit'], 'NT frac', d['roofline']['frac'])"; }
{
run "default (side by side)" "A=1" ""
run "NT_TILE=3 (phased 256^2 NT everywhere)" "ST5_NT_TILE=3" ""
run "TN_PHASED=1" "ST5_TN_PHASED=1" ""
run "NT_SLOTS=4" "ST5_NT_SLOTS=4" ""
run "SPLITK_TARGET=256" "ST5_SPLITK_TARGET=256" ""
run "SPLITK_TARGET=512" "ST5_SPLITK_TARGET=512" ""
run "default again" "A=1" ""
run "in turn" "A=1" "--micro in_turn"
run "batch 32 side by side" "A=1" "--batch 32 --steps 10"
run "batch 32 in turn" "A=1" "--batch 32 --steps 10 --micro in_turn"
} > $O/knobs.log 2>&1
cat $O/knobs.log
R=$PWD
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --steps 13 --warmup 5 --no-cpu-baseline > /tmp/kt.log 2>&1 < /dev/null; echo "rocprof rc=$?"
  f=$(ls /tmp/kt/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $R/$O/sbs_kernel_stats.csv; tail -2 /tmp/kt.log | cut -c1-200 )

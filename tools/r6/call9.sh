#!/bin/bash
mkdir -p gpurun_out/r6
for v in "in_turn phased" "side_by_side one_message" "side_by_side phased"; do
  set -- $v
  timeout 400 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --micro $1 --exchange $2 > gpurun_out/r6/two_$1_$2.json 2> gpurun_out/r6/two_$1_$2.err
  python -c "import json;d=json.load(open('gpurun_out/r6/two_$1_$2.json'));print('2 ranks on one GPU, $1 $2:', d['ms_per_step'], 'ms; local', d['config']['exchange']['local_phase_ms'])" || tail -8 gpurun_out/r6/two_$1_$2.err
done

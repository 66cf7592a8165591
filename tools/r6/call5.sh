#!/bin/bash
# whole-update A/B: the text micro-batch's weight-gradient groups on the speech stream (ST5_WGRAD_MOVE)
mkdir -p gpurun_out/r6
for rep in 1 2; do
for mv in 0 1; do
  ST5_WGRAD_MOVE=$mv timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r6/ab_move_${mv}_$rep.json 2> gpurun_out/r6/ab_move_${mv}_$rep.err
  python -c "import json;d=json.load(open('gpurun_out/r6/ab_move_${mv}_$rep.json'));print('wgrad move $mv rep $rep:', d['ms_per_step'], 'ms')" || tail -5 gpurun_out/r6/ab_move_${mv}_$rep.err
done; done
timeout 900 python -m pytest tests/test_bench_update_gpu.py -x -q 2>&1 | tail -5

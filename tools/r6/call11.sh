#!/bin/bash
# GPU_MAX_HW_QUEUES (ROCm default 4): does the number of hardware queues limit the concurrency of a replayed update?
mkdir -p gpurun_out/r6
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r6/q_$tag.json 2> gpurun_out/r6/q_$tag.err
  python -c "import json;d=json.load(open('gpurun_out/r6/q_$tag.json'));print('$tag:', d['ms_per_step'], 'ms')" || tail -3 gpurun_out/r6/q_$tag.err; }
for rep in 1 2; do
run base_$rep A=1
run q8_$rep GPU_MAX_HW_QUEUES=8
run q2_$rep GPU_MAX_HW_QUEUES=2
run q8_move_$rep GPU_MAX_HW_QUEUES=8 ST5_WGRAD_MOVE=1
run q8_third_chain_$rep GPU_MAX_HW_QUEUES=8 ST5_WGRAD_STREAM=1 ST5_SBS_OWNER=1
done

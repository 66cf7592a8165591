#!/bin/bash
mkdir -p gpurun_out/r6
ST5_PHASED_DEFER_EXCHANGE=1 timeout 400 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r6/two_defer.json 2> gpurun_out/r6/two_defer.err
python -c "import json;d=json.load(open('gpurun_out/r6/two_defer.json'));print('2 ranks on one GPU, sbs phased graphs, exchange deferred to the end:', d['ms_per_step'], 'ms; local', d['config']['exchange']['local_phase_ms'])" || tail -8 gpurun_out/r6/two_defer.err
GPU_MAX_HW_QUEUES=8 timeout 400 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r6/two_q8.json 2> gpurun_out/r6/two_q8.err
python -c "import json;d=json.load(open('gpurun_out/r6/two_q8.json'));print('2 ranks on one GPU, sbs phased, GPU_MAX_HW_QUEUES=8:', d['ms_per_step'], 'ms; local', d['config']['exchange']['local_phase_ms'])" || tail -8 gpurun_out/r6/two_q8.err

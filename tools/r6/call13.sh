#!/bin/bash
mkdir -p gpurun_out/r6
for q in 4 8 4 8; do
  GPU_MAX_HW_QUEUES=$q ST5_DDP_FORCE_COLLECTIVES=1 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --exchange phased > gpurun_out/r6/fq_$q.json 2> gpurun_out/r6/fq_$q.err
  python -c "import json;d=json.load(open('gpurun_out/r6/fq_$q.json'));e=d['config'].get('exchange',{});print('one-rank RCCL group, phased, $q hardware queues:', d['ms_per_step'], 'ms; local phase', e.get('local_phase_ms'))" || tail -5 gpurun_out/r6/fq_$q.err
done

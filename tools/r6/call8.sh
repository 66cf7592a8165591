#!/bin/bash
# what the phased side-by-side exchange costs on one GPU (one-rank RCCL group: the collectives are issued, nothing travels), and the 2-rank
# shared-GPU functional line of the default form
mkdir -p gpurun_out/r6
for ex in one_message phased one_message phased; do
  ST5_DDP_FORCE_COLLECTIVES=1 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --exchange $ex > gpurun_out/r6/force_$ex.json 2> gpurun_out/r6/force_$ex.err
  python -c "import json;d=json.load(open('gpurun_out/r6/force_$ex.json'));e=d['config'].get('exchange',{});print('one-rank RCCL group, $ex:', d['ms_per_step'], 'ms; form', e.get('form'), 'messages', e.get('message_bytes'), 'local phase', e.get('local_phase_ms'))" || tail -5 gpurun_out/r6/force_$ex.err
done
timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r6/bench_2rank_shared_gpu.json 2> gpurun_out/r6/bench_2rank_shared_gpu.err
python -c "import json;d=json.load(open('gpurun_out/r6/bench_2rank_shared_gpu.json'));print('2 ranks on one GPU:', d['ms_per_step'], d['config']['exchange'], d['config'].get('note'))" || tail -8 gpurun_out/r6/bench_2rank_shared_gpu.err

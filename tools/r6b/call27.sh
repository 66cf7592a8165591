#!/bin/bash
# several-rank forms at the closing tree: two processes sharing the one GPU (gloo; functional), and the one-rank RCCL group arms
mkdir -p gpurun_out/r6b
timeout 900 python bench.py --gpus 2 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r6_bench_2rank_shared_gpu.json 2> gpurun_out/r6b/bench_2rank.err < /dev/null; echo "2-rank rc=$?"; head -c 700 gpurun_out/r6_bench_2rank_shared_gpu.json; echo
for ex in phased one_message; do
  ST5_DDP_FORCE_COLLECTIVES=1 NCCL_ALGO=Ring timeout 400 python bench.py --exchange $ex --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r6_force_$ex.json 2> gpurun_out/r6b/force_$ex.err < /dev/null
  python -c "import json;d=json.load(open('gpurun_out/r6_force_$ex.json'));print('one-rank RCCL group, $ex:', d['ms_per_step'], 'ms', d['config'].get('exchange'))"
done
for rep in 1 2 3; do
for lk in 0 64; do
  ST5_NT_LONGK=$lk,36 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r6b/base_longk${lk}_$rep.json 2> gpurun_out/r6b/base_longk${lk}_$rep.err < /dev/null
  python -c "import json;d=json.load(open('gpurun_out/r6b/base_longk${lk}_$rep.json'));print('base N=768 long-K shapes on the phased NT kernel (min tiles $lk) rep $rep:', d['ms_per_step'], 'ms')"
done; done

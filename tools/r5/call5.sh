#!/bin/bash
# GPU call 5 of round 5: new tests (B = 32 equality, two ranks side by side), the per-GPU shape of cfg 4 / cfg 5 (B = 32; Large bf16 vs
# fp8 alternating on one box), cfg 3, split-K sweep in the side-by-side mode, the several-rank line on one GPU (forced collectives).
O=gpurun_out/r5e; mkdir -p $O
timeout 1500 python -m pytest tests/test_bench_update_gpu.py::test_benched_update_at_the_cfg4_per_gpu_batch tests/test_two_rank_gpu.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
one() { env $2 timeout 400 python bench.py --no-cpu-baseline $3 > $O/$1.json 2> $O/$1.err; python -c "import json; d=json.load(open('$O/$1.json')); print('$1:', d['ms_per_step'], 'ms', d['value'], d['unit'], 'NT frac', d['roofline']['frac'], d['config'].get('exchange'))"; }
{
one base_b32 A=1 "--batch 32 --steps 15 --warmup 5"
one large_b32_bf16 A=1 "--arch large --batch 32 --steps 10 --warmup 5"
one large_b32_fp8 A=1 "--arch large --batch 32 --steps 10 --warmup 5 --dtype fp8"
one large_b32_bf16_again A=1 "--arch large --batch 32 --steps 10 --warmup 5"
one large_b32_fp8_again A=1 "--arch large --batch 32 --steps 10 --warmup 5 --dtype fp8"
one large_b8_bf16 A=1 "--arch large --steps 15 --warmup 5"
one large_b8_fp8 A=1 "--arch large --steps 15 --warmup 5 --dtype fp8"
one splitk128 ST5_SPLITK_TARGET=128 "--steps 30 --warmup 5"
one splitk192 ST5_SPLITK_TARGET=192 "--steps 30 --warmup 5"
one base_default A=1 "--steps 30 --warmup 5"
one forced_collectives "ST5_DDP_FORCE_COLLECTIVES=1 MASTER_PORT=29533" "--steps 20 --warmup 5"
timeout 300 python bench.py --config 3 --steps 20 --warmup 5 > $O/cfg3.json 2> $O/cfg3.err; python -c "import json; d=json.load(open('$O/cfg3.json')); print('cfg3:', d['ms_per_step'], 'ms', d['value'], d['unit'], d['vocoder'])" 2>&1 | cut -c1-400
} > $O/lines.log 2>&1
cat $O/lines.log

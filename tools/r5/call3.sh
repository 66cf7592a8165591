#!/bin/bash
# GPU call 3 of round 5: with the lgkmcnt(0) fix in fa2::bwd_dkv_kernel -- (A) is every concurrent mode of the update now bit-identical
# to the in-turn update over 300 replays?  (B) the two-kernel reproducer, fixed library vs the round-4 form; (C) what the modes are
# worth; (D) the tests this round changed.
O=gpurun_out/r5c; mkdir -p $O
H="python tools/r5/replay_hunt.py"; G='HUNT|DIFF|Error|error|assert|differ|identical'
{
timeout 200 $H run $O/ref.json --n 300 2>&1 | grep -E "$G"
for i in 1 2 3; do timeout 200 $H run $O/sbs$i.json --n 300 --mode side_by_side 2>&1 | grep -E "$G"; $H diff $O/ref.json $O/sbs$i.json; done
for i in 1 2; do timeout 200 $H run $O/sbsw$i.json --n 300 --mode side_by_side --wgrad 1 2>&1 | grep -E "$G"; $H diff $O/ref.json $O/sbsw$i.json; done
timeout 200 $H run $O/itw.json --n 300 --wgrad 1 2>&1 | grep -E "$G"; $H diff $O/ref.json $O/itw.json
timeout 200 $H run $O/it2.json --n 300 --mode in_turn_2buf 2>&1 | grep -E "$G"; $H diff $O/ref.json $O/it2.json
echo "round-4 form of the dkv barrier (libspeecht5_hip_nolgkm.so):"
ST5_HIP_LIB=$PWD/speecht5_amd/libspeecht5_hip_nolgkm.so timeout 200 $H run $O/sbs_old.json --n 60 --mode side_by_side 2>&1 | grep -E "$G"; $H diff $O/ref.json $O/sbs_old.json
} > $O/A.log 2>&1
P="python tools/r5/dkv_pair.py"; F='PAIR|Error|error|assert'
{
for ga in "0 attn" "1 attn" "0 self" "0 fwd" "0 bwdk"; do set -- $ga; timeout 120 $P $2 --graph $1 --reps 1600 2>&1 | grep -E "$F"; done
ST5_HIP_LIB=$PWD/speecht5_amd/libspeecht5_hip_nolgkm.so timeout 120 $P attn --reps 1600 2>&1 | grep -E "$F"
} > $O/B.log 2>&1
{
for m in "in_turn 0" "side_by_side 0" "side_by_side 1" "in_turn 1" "in_turn 0" "side_by_side 0"; do set -- $m
  ST5_WGRAD_STREAM=$2 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --micro $1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 wgrad $2:', d['ms_per_step'], 'ms', d['value'], d['unit'])"
done
} > $O/C.log 2>&1
timeout 900 python -m pytest tests/test_flash_gpu.py tests/test_fp8_gpu.py tests/test_bench_update_gpu.py tests/test_two_rank_gpu.py tests/test_bf16_path_gpu.py -x -q -m gpu > $O/D.log 2>&1
cat $O/A.log $O/B.log $O/C.log; tail -15 $O/D.log

#!/bin/bash
# GPU call 1 of round 5: (A) in-turn replay hunt arms, (B) side-by-side discriminator arms on the whole update, (C) the two-kernel
# reproducer, (D) a bench line for this box.  Everything lands in gpurun_out/r5a/.
O=gpurun_out/r5a; mkdir -p $O
H="python tools/r5/replay_hunt.py"
{
echo "## (A) in-turn replay: with / without per-update synchronisation"
timeout 200 $H run $O/a_nosync1.json --n 300 --sync 0 --dump-at 194,195,196 2>&1 | grep -E "HUNT|Error|error|assert"
timeout 200 $H run $O/a_nosync2.json --n 300 --sync 0 2>&1 | grep -E "HUNT|Error|error|assert"
timeout 200 $H run $O/a_sync1.json   --n 300 --sync 1 --dump-at 194,195,196 2>&1 | grep -E "HUNT|Error|error|assert"
timeout 200 $H run $O/a_sync2.json   --n 300 --sync 1 2>&1 | grep -E "HUNT|Error|error|assert"
timeout 200 $H run $O/a_sleep.json   --n 300 --sync 1 --sleep-ms 20 2>&1 | grep -E "HUNT|Error|error|assert"
timeout 200 $H run $O/a_events.json  --n 300 --sync 1 --extra-events 20 2>&1 | grep -E "HUNT|Error|error|assert"
timeout 200 $H run $O/a_seed_nosync.json --n 300 --sync 0 --seed-off 1000 2>&1 | grep -E "HUNT|Error|error|assert"
timeout 200 $H run $O/a_seed_sync.json   --n 300 --sync 1 --seed-off 1000 2>&1 | grep -E "HUNT|Error|error|assert"
for p in "a_nosync1 a_nosync2" "a_nosync1 a_sync1" "a_nosync2 a_sync2" "a_sync1 a_sync2" "a_nosync2 a_sleep" "a_nosync2 a_events" "a_seed_nosync a_seed_sync"; do
  set -- $p; $H diff $O/$1.json $O/$2.json
done
} > $O/A.log 2>&1
{
echo "## (B) side-by-side replay vs the in-turn trajectory (40 updates), one library variant per arm"
S="python tools/r4/sbs_hunt.py"
timeout 200 $S record in_turn 40 0 $O/b_ref.json 2>&1 | grep -E "HUNT|Error|error"
for v in base dmanop sleep fat base; do
  L=$PWD/speecht5_amd/libspeecht5_hip_$v.so; [ $v = base ] && L=$PWD/speecht5_amd/libspeecht5_hip.so
  echo "arm $v"; ST5_HIP_LIB=$L timeout 200 $S check side_by_side 40 0 $O/b_ref.json $O/b_bad_$v.json 2>&1 | grep -E "HUNT|Error|error"
done
echo "arm abl1 (no window DMA; its own in-turn reference)"
ST5_HIP_LIB=$PWD/speecht5_amd/libspeecht5_hip_abl1.so timeout 200 $S record in_turn 40 0 $O/b_ref_abl1.json 2>&1 | grep -E "HUNT|Error|error"
ST5_HIP_LIB=$PWD/speecht5_amd/libspeecht5_hip_abl1.so timeout 200 $S check side_by_side 40 0 $O/b_ref_abl1.json $O/b_bad_abl1.json 2>&1 | grep -E "HUNT|Error|error"
} > $O/B.log 2>&1
{
echo "## (C) two-kernel reproducer: attention backward (speech shape, bias + dropout) beside an aggressor on a second stream"
P="python tools/r5/dkv_pair.py"
for ga in "0 none" "0 nt" "0 attn" "1 none" "1 nt" "1 nt_s" "1 tn" "1 attn" "1 self"; do set -- $ga; g=$1; ag=$2;
  timeout 120 $P $ag --graph $g --reps 1600 2>&1 | grep -E "PAIR|wrong launch|Error|error|assert" | head -8
done
for v in dmanop sleep fat abl1; do
  ST5_HIP_LIB=$PWD/speecht5_amd/libspeecht5_hip_$v.so timeout 120 $P nt --graph 1 --reps 1600 2>&1 | grep -E "PAIR|wrong launch|Error|error|assert" | head -4
  ST5_HIP_LIB=$PWD/speecht5_amd/libspeecht5_hip_$v.so timeout 120 $P attn --graph 1 --reps 1600 2>&1 | grep -E "PAIR|wrong launch|Error|error|assert" | head -4
done
} > $O/C.log 2>&1
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -3 $O/A.log; tail -3 $O/B.log; tail -3 $O/C.log; cat $O/bench.json | cut -c1-300

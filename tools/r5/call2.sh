#!/bin/bash
# GPU call 2 of round 5: isolate the aggressor and the victim-side mechanism of the attention-backward difference (tools/r5/dkv_pair.py),
# and re-run round 4's own in-turn hunt tool (does its update-195 event exist on this box?).
O=gpurun_out/r5b; mkdir -p $O
P="python tools/r5/dkv_pair.py"
F='PAIR|wrong launch|dK \(b|first group|Error|error|assert'
{
echo "## aggressor isolation (victim: full attention backward, bias + dropout; eager, two streams)"
for ag in fwd qpt bwdk dqpgemm; do timeout 120 $P $ag --reps 800 2>&1 | grep -E "$F" | head -5; done
echo "## aggressor variants"
timeout 120 $P attn --abias 0 --reps 800 2>&1 | grep -E "$F" | head -5
timeout 120 $P attn --apdrop 0 --reps 800 2>&1 | grep -E "$F" | head -5
echo "## victim variants (aggressor: attn)"
timeout 120 $P attn --bias 0 --reps 800 2>&1 | grep -E "$F" | head -5
timeout 120 $P attn --pdrop 0 --reps 800 2>&1 | grep -E "$F" | head -5
echo "## library variants (aggressor: attn)"
for v in lgkm abl2 abl3 abl4; do
  ST5_HIP_LIB=$PWD/speecht5_amd/libspeecht5_hip_$v.so timeout 120 $P attn --reps 800 2>&1 | grep -E "$F" | head -5
done
} > $O/C.log 2>&1
bash tools/r4/inturn_hunt.sh r5b_inturn 300 0 2 > $O/A.log 2>&1
cat $O/C.log | grep -E "PAIR|##"; cat $O/A.log

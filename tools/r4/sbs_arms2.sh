#!/bin/bash
# second discriminator round: the dkv kernel's data 16 KB into its LDS allocation (libspeecht5_hip_pad.so: -DFA2_DKV_HEAD_PAD) -- does a
# neighbour block on the CU write past ITS end into the start of the dkv block's LDS?
O=gpurun_out/${1:-arms2}; mkdir -p $O; N=${2:-190}
timeout 200 python tools/r4/sbs_hunt.py record in_turn $N 0 $O/ref.json 2>&1 | grep -E "HUNT|Error|error"
echo "arm pad";  ST5_HIP_LIB=$PWD/speecht5_amd/libspeecht5_hip_pad.so timeout 200 python tools/r4/sbs_hunt.py check side_by_side $N 0 $O/ref.json $O/bad_pad.json 2>&1 | grep -E "HUNT|Error|error"
echo "arm base"; timeout 200 python tools/r4/sbs_hunt.py check side_by_side $N 0 $O/ref.json $O/bad_base.json 2>&1 | grep -E "HUNT|Error|error"

#!/bin/bash
# discriminators for the side-by-side replay difference: one in-turn reference trajectory, then the replayed
# side-by-side update checked against it  (a) as shipped,  (b) on the first-generation attention kernels,  (c) with one fat dkv block
# per CU (libspeecht5_hip_fat.so from tools/r4/fa_variant_build.sh fat -DFA2_DKV_FAT_LDS).  "clean" in an arm that the baseline
# fails says where to look.
O=gpurun_out/${1:-arms}; mkdir -p $O; N=${2:-300}
timeout 200 python tools/r4/sbs_hunt.py record in_turn $N 0 $O/ref.json 2>&1 | grep -E "HUNT|Error|error"
echo "arm base";  timeout 200 python tools/r4/sbs_hunt.py check side_by_side $N 0 $O/ref.json $O/bad_base.json 2>&1 | grep -E "HUNT|Error|error"
echo "arm impl1"; HUNT_FLASH_IMPL=1 timeout 200 python tools/r4/sbs_hunt.py check side_by_side $N 0 $O/ref.json $O/bad_impl1.json 2>&1 | grep -E "HUNT|Error|error"
echo "arm fat";   ST5_HIP_LIB=$PWD/speecht5_amd/libspeecht5_hip_fat.so timeout 200 python tools/r4/sbs_hunt.py check side_by_side $N 0 $O/ref.json $O/bad_fat.json 2>&1 | grep -E "HUNT|Error|error"
echo "arm base again"; timeout 200 python tools/r4/sbs_hunt.py check side_by_side $N 0 $O/ref.json $O/bad_base2.json 2>&1 | grep -E "HUNT|Error|error"

#!/bin/bash
# third discriminator: every fa2 kernel declares 256 VGPRs (libspeecht5_hip_v256.so: -DFA2_PAD256), like the 128^2 GEMM kernels
O=gpurun_out/${1:-arms3}; mkdir -p $O; N=${2:-190}
timeout 200 python tools/r4/sbs_hunt.py record in_turn $N 0 $O/ref.json 2>&1 | grep -E "HUNT|Error|error"
echo "arm v256";  ST5_HIP_LIB=$PWD/speecht5_amd/libspeecht5_hip_v256.so timeout 200 python tools/r4/sbs_hunt.py check side_by_side $N 0 $O/ref.json $O/bad_v256.json 2>&1 | grep -E "HUNT|Error|error"

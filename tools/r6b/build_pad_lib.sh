#!/bin/bash
# speecht5_amd/libspeecht5_hip_pad.so: the library WITH ST5_PAD_TO_256_VGPRS (round 3-5 default: the 128x128 GEMM / attention kernels declare
# 256 registers, so that no third wave of any kernel shares their SIMDs).  A/B timing: ST5_HIP_LIB=<path>.
set -e
cd "$(dirname "$0")/../../speecht5_amd/csrc"
mkdir -p build_pad
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-comment -Xclang -target-feature -Xclang -packed-fp32-ops -DST5_PAD256"
for f in gemm flash_attn2; do /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o build_pad/$f.o 2> >(grep -v "not a recognized feature" >&2) & done; wait
OBJS=""
for f in norm softmax conv0 elementwise optim flash_attn runtime batchnorm ctc_prefix losses vq nce ctc_loss conv1d_narrow; do OBJS="$OBJS build/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_pad/gemm.o build_pad/flash_attn2.o $OBJS -o ../libspeecht5_hip_pad.so
echo built libspeecht5_hip_pad.so

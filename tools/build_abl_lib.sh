#!/bin/bash
# Builds speecht5_amd/libspeecht5_hip_abl<N>.so: the library with -DGEMM_ABL=<N> in gemm.hip (2 = no LDS-DMA loads in the
# k-loop, 3 = no MFMAs, 1 = no epilogue).  Timing experiments only (results are wrong by construction).
set -e
N=$1
cd "$(dirname "$0")/../speecht5_amd/csrc"
mkdir -p build_abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-comment -DGEMM_ABL=$N -c gemm.hip -o build_abl/gemm$N.o
OBJS="build/norm.o build/softmax.o build/conv0.o build/elementwise.o build/optim.o build/flash_attn.o build/runtime.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_abl/gemm$N.o $OBJS -o ../libspeecht5_hip_abl$N.so
echo built libspeecht5_hip_abl$N.so

#!/bin/bash
# builds speecht5_amd/libspeecht5_hip_<tag>.so: the library with extra -D flags in gemm.hip (A/B timing only; run with ST5_HIP_LIB=<path>)
# usage: gemm_variant_build.sh <tag> <flags...>     e.g.  gemm_variant_build.sh g0 -DGEMM_GROUP_N=0
set -e
cd "$(dirname "$0")/../../speecht5_amd/csrc"
TAG=$1; shift
mkdir -p build_abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-comment -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -c gemm.hip -o build_abl/gemm_$TAG.o 2> >(grep -v "not a recognized feature" >&2)
OBJS=$(ls build/*.o | grep -v "build/gemm.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_abl/gemm_$TAG.o $OBJS -o ../libspeecht5_hip_$TAG.so
echo built libspeecht5_hip_$TAG.so

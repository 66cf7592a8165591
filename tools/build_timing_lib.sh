#!/bin/bash
# Builds speecht5_amd/libspeecht5_hip_timing.so (git-ignored): the library with -DGEMM_TIMING (s_memtime phase probes in the NT kernels).
# Every other object comes from the normal build (speecht5_amd/csrc/build).  Use: ST5_HIP_LIB=<path> python tools/gemm_timing.py
set -e
cd "$(dirname "$0")/../speecht5_amd/csrc"
make -j8 > /dev/null
mkdir -p build_timing
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-comment -Xclang -target-feature -Xclang -packed-fp32-ops -DGEMM_TIMING \
    -c gemm.hip -o build_timing/gemm.o 2> >(grep -v "not a recognized feature for this target" >&2)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_timing/gemm.o $(ls build/*.o | grep -v "build/gemm.o") -o ../libspeecht5_hip_timing.so
echo built speecht5_amd/libspeecht5_hip_timing.so

#!/bin/bash
# Builds speecht5_amd/libspeecht5_hip_timing.so (git-ignored): the library with -DGEMM_TIMING (s_memtime phase probes in the NT kernels).
# Use it with ST5_HIP_LIB=<path> python tools/gemm_timing.py
set -e
cd "$(dirname "$0")/../speecht5_amd/csrc"
mkdir -p build_timing ../../gpurun_out
for f in gemm norm softmax conv0 elementwise optim flash_attn; do
  if [ $f = gemm ]; then X="-DGEMM_TIMING"; else X=""; fi
  if [ $f = gemm ] || [ ! -f build_timing/$f.o ] || [ $f.hip -nt build_timing/$f.o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-comment $X -c $f.hip -o build_timing/$f.o
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_timing/*.o -o ../libspeecht5_hip_timing.so
echo built speecht5_amd/libspeecht5_hip_timing.so

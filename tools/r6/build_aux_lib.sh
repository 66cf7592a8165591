#!/bin/bash
# speecht5_amd/libspeecht5_hip_aux<A>_<B>.so: the library with the 128^2 NT kernel's operand LDS-DMA loads under another cache policy
# (A/B measurements: ST5_HIP_LIB=<path> python bench.py).  Needs the normal build's objects.
set -e
A=$1; B=$2
cd "$(dirname "$0")/../../speecht5_amd/csrc"
mkdir -p build_abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-comment -Xclang -target-feature -Xclang -packed-fp32-ops -DGLDS_AUX_A=$A -DGLDS_AUX_B=$B -c gemm.hip -o build_abl/gemm_aux${A}_$B.o 2> >(grep -v "not a recognized feature" >&2)
OBJS=$(ls build/*.o | grep -v "build/gemm.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_abl/gemm_aux${A}_$B.o $OBJS -o ../libspeecht5_hip_aux${A}_$B.so
echo built libspeecht5_hip_aux${A}_$B.so

#!/bin/bash
# builds speecht5_amd/libspeecht5_hip_fa<N>.so: the library with -DFA2_ABL=<N> in flash_attn2.hip (timing experiments only)
set -e
cd "$(dirname "$0")/../../speecht5_amd/csrc"
mkdir -p build_abl
for N in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-comment -Xclang -target-feature -Xclang -packed-fp32-ops -DFA2_ABL=$N -c flash_attn2.hip -o build_abl/fa2_$N.o 2> >(grep -v "not a recognized feature" >&2)
  OBJS=$(ls build/*.o | grep -v flash_attn2.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_abl/fa2_$N.o $OBJS -o ../libspeecht5_hip_fa$N.so
  echo built libspeecht5_hip_fa$N.so
done

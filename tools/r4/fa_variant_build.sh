#!/bin/bash
# builds speecht5_amd/libspeecht5_hip_<tag>.so: the library with extra -D flags in flash_attn2.hip (A/B / discriminator builds only)
# usage: fa_variant_build.sh <tag> <flags...>      e.g.  fa_variant_build.sh fat -DFA2_DKV_FAT_LDS
set -e
cd "$(dirname "$0")/../../speecht5_amd/csrc"
TAG=$1; shift
mkdir -p build_abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-comment -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -c flash_attn2.hip -o build_abl/fa2_$TAG.o 2> >(grep -v "not a recognized feature" >&2)
OBJS=$(ls build/*.o | grep -v flash_attn2.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_abl/fa2_$TAG.o $OBJS -o ../libspeecht5_hip_$TAG.so
echo built libspeecht5_hip_$TAG.so

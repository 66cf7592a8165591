#!/bin/bash
# builds tools/diag/libst5_diag.so (the st5_debug_* canary kernels; not part of the product library)
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared canary.hip -o libst5_diag.so
echo built tools/diag/libst5_diag.so

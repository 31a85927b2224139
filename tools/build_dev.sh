#!/bin/bash
# Development build of the device library: the experiment switches (prim::dev_env, GRLBWT_DEV_*) are compiled in, the
# register / scratch / LDS report of every kernel is kept beside it (tools/kernel_resources.py reads it).
#   tools/build_dev.sh            -> tools/_build/libgrlbwt_dev.so, tools/_build/kernel_resources.txt
#   GRLBWT_HIP_LIB=tools/_build/libgrlbwt_dev.so GRLBWT_DEV_...=... python bench.py ...
# The product library (grlbwt_amd/csrc/libgrlbwt_hip.so, __graft_entry__.build_hip) is built WITHOUT the macro and reads none of them.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$R/tools/_build"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -DGRLBWT_DEV_SWITCHES \
    -Rpass-analysis=kernel-resource-usage -o "$R/tools/_build/libgrlbwt_dev.so" "$R/grlbwt_amd/csrc/engine_hip.hip" -lz \
    2> "$R/tools/_build/kernel_resources.txt" || { grep -v "remark:" "$R/tools/_build/kernel_resources.txt" | head -50; exit 1; }
grep -c "Function Name" "$R/tools/_build/kernel_resources.txt"

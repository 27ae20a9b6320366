#!/bin/bash
# experiment build of the library with the MV3D_TUNING environment hooks (tools/ probes only): build_variants/libmv3d_tuning.so
set -e
cd "$(dirname "$0")/.."
mkdir -p build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -pthread -Wall -Wno-unused-function \
    -DMV3D_TUNING $MV3D_EXTRA_FLAGS mv3d_tf_amd/csrc/*.hip -o build_variants/${MV3D_TUNING_OUT:-libmv3d_tuning.so}
ls -la build_variants/${MV3D_TUNING_OUT:-libmv3d_tuning.so}

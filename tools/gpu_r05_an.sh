#!/bin/bash
# GPU box, round 5: the input layer (conv1_1, 256 x 64 tiles, write-bound at 2.1 TB/s): smaller tiles / three stages
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05an; mkdir -p $OUT
L=build_variants/libmv3d_tuning.so
{
for r in 1 2; do for f in 0 1 2 3 4; do echo "== MV3D_CONV_FIRST=$f run $r"; MV3D_CONV_FIRST=$f timeout 300 python tools/conv_probe.py 16 --no-torch --lib $L --only conv1_1 2>&1 | grep -v amdgpu.ids | tail -6; done; done
} 2>&1 | tee $OUT/conv_first_ab.txt

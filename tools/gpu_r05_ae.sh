#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05ae; mkdir -p $OUT
L=build_variants/libmv3d_tuning.so
{
for w in 16 64; do for d in 0 4; do echo "== W=$w DBG=$d"; PAIR_ONLY=1 NB=8 ROUNDS=4 MV3D_IDX_DBG=1 MV3D_RGT_W=$w MV3D_RGT_DBG=$d timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair " | tail -1; done; done
echo "== W=64 ORDER=1"; PAIR_ONLY=1 NB=8 ROUNDS=4 MV3D_RGT_W=64 MV3D_RGT_ORDER=1 timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair " | tail -1
echo "== W=64 fv 2x2"; PAIR_ONLY=1 NB=8 ROUNDS=4 MV3D_RGT_W=64 MV3D_RGT_PX=$((4 + (16<<8) + (16<<16))) timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair " | tail -1
echo "== trace W=64"; MV3D_RGT_W=64 timeout 200 python tools/roi_tiles_trace.py --lib $L 2>&1 | grep -v amdgpu.ids
} 2>&1 | tee $OUT/tiles_w64.txt

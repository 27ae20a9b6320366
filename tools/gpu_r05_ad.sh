#!/bin/bash
# GPU box, round 5: where the tile RoiPoolGrad's time goes -- phases switched off (MV3D_RGT_DBG bits: 1 no expansion, 2 no drains,
# 4 loads but no ordered adds, 8 no hits) and the per-wave stamps
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05ad; mkdir -p $OUT
L=build_variants/libmv3d_tuning.so
{
for d in 0 4 2 1 8; do echo "== DBG=$d"; PAIR_ONLY=1 NB=8 ROUNDS=4 MV3D_IDX_DBG=1 MV3D_RGT_DBG=$d timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair "; done
echo "== trace"; timeout 200 python tools/roi_tiles_trace.py --lib $L 2>&1 | grep -v amdgpu.ids
} 2>&1 | tee $OUT/tiles_ablate.txt

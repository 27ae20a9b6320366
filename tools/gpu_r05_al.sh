#!/bin/bash
# GPU box, round 5: tile RoiPoolGrad v4' (list + three rotating buffers, branch-free two-row target)
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05al; mkdir -p $OUT
{
PXA=$((4 + (8<<8) + (16<<16))); PXC=$((2 + (16<<8) + (16<<16))); PXB=$((2 + (8<<8) + (8<<16)))
for cfg in "w4 8" "w2 16"; do
  set -- $cfg; L=build_variants/libmv3d_tuning_$1.so; W=$2
  echo "==== lib $1 W=$W"
  echo "== check tiles vs plain"; MV3D_PAIR_TILES=1 MV3D_RGT_W=$W NB=4 ROUNDS=2 timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "identical\|differ\|Error\|error"
  for px in $PXA $PXC $PXB; do echo "== tiles PX=$(printf %x $px)"; PAIR_ONLY=1 MV3D_PAIR_TILES=1 MV3D_RGT_W=$W MV3D_RGT_PX=$px timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair " | tail -1; done
  echo "== trace"; MV3D_PAIR_TILES=1 MV3D_RGT_W=$W MV3D_RGT_PX=$PXC timeout 200 python tools/roi_tiles_trace.py --lib $L 2>&1 | grep -v "amdgpu.ids\|expand + drains\|last drain"
done
} 2>&1 | tee $OUT/tiles_v4b.txt

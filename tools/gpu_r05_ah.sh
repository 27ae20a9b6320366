#!/bin/bash
# GPU box, round 5: tile RoiPoolGrad v4 (entry list per round, three rotating load buffers): parity, A / B, stamps
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05ah; mkdir -p $OUT
{
for cfg in "w4 8" "w2 16" "w3 8"; do
  set -- $cfg; L=build_variants/libmv3d_tuning_$1.so; W=$2
  echo "==== lib $1 W=$W"
  echo "== check tiles vs plain"; MV3D_PAIR_TILES=1 MV3D_RGT_W=$W NB=4 ROUNDS=2 timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "identical\|differ\|Error\|error"
  for o in 0 1; do echo "== tiles ORDER=$o"; PAIR_ONLY=1 MV3D_PAIR_TILES=1 MV3D_RGT_W=$W MV3D_RGT_ORDER=$o timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair " | tail -1; done
  echo "== trace"; MV3D_PAIR_TILES=1 MV3D_RGT_W=$W MV3D_RGT_ORDER=1 timeout 200 python tools/roi_tiles_trace.py --lib $L 2>&1 | grep -v amdgpu.ids
done
} 2>&1 | tee $OUT/tiles_v4.txt

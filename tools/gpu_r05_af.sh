#!/bin/bash
# GPU box, round 5: tile RoiPoolGrad v2 (register accumulators, 8-byte entries, lane = hit column test): parity, A / B, ablation, stamps
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05af; mkdir -p $OUT
L=build_variants/libmv3d_tuning.so
{
echo "== pair tests on the tile kernel (tuning lib, MV3D_PAIR_TILES=1)"
cp mv3d_tf_amd/libmv3d_hip.so /tmp/ship.so; cp $L mv3d_tf_amd/libmv3d_hip.so; MV3D_PAIR_TILES=1 timeout 900 python -m pytest tests/test_roi_pair.py tests/test_roipool_pin.py -x -q -m gpu 2>&1 | tail -5; cp /tmp/ship.so mv3d_tf_amd/libmv3d_hip.so
echo "== check tiles vs plain"; MV3D_PAIR_TILES=1 NB=4 ROUNDS=2 timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "identical\|differ\|Error\|error"
for r in 1 2; do
  echo "== old (index + gather) run $r"; PAIR_ONLY=1 MV3D_PAIR_TILES=0 timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair " | tail -1
  echo "== tiles run $r"; PAIR_ONLY=1 MV3D_PAIR_TILES=1 timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair " | tail -1
  echo "== tiles ORDER=1 run $r"; PAIR_ONLY=1 MV3D_PAIR_TILES=1 MV3D_RGT_ORDER=1 timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair " | tail -1
done
for d in 4 2 1 8; do echo "== DBG=$d"; PAIR_ONLY=1 NB=8 ROUNDS=4 MV3D_PAIR_TILES=1 MV3D_IDX_DBG=1 MV3D_RGT_DBG=$d timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair " | tail -1; done
for px in $((1 + (16<<8) + (16<<16))) $((4 + (16<<8) + (16<<16))) $((2 + (8<<8) + (8<<16))) $((2 + (4<<8) + (4<<16))) $((2 + (16<<8) + (8<<16))); do
  echo "== tiles PX=$(printf %x $px)"; PAIR_ONLY=1 MV3D_PAIR_TILES=1 MV3D_RGT_PX=$px timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair " | tail -1
done
echo "== trace"; MV3D_PAIR_TILES=1 timeout 200 python tools/roi_tiles_trace.py --lib $L 2>&1 | grep -v amdgpu.ids
} 2>&1 | tee $OUT/tiles_v2.txt

#!/bin/bash
# GPU box, round 5: RoiPoolGrad of the pair as one launch of LDS tiles (roi_grad_tiles.hip) -- parity tests, then A / B against the
# three-launch index + gather (MV3D_PAIR_TILES=0) with tile-size and loads-in-flight sweeps
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05ac; mkdir -p $OUT
L=build_variants/libmv3d_tuning.so
timeout 900 python -m pytest tests/test_roi_pair.py -x -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest_pair.txt
{
echo "== check tiles vs plain"; MV3D_PAIR_TILES=1 NB=4 ROUNDS=2 timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "identical\|differ\|Error\|error"
for r in 1 2; do
  echo "== old (index + gather) run $r"; PAIR_ONLY=1 MV3D_PAIR_TILES=0 timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair "
  for w in 8 16 32; do echo "== tiles W=$w run $r"; PAIR_ONLY=1 MV3D_PAIR_TILES=1 MV3D_RGT_W=$w timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair "; done
done
# tile pixels: fv | bev << 8 | rgb << 16
for px in $((2 + (16<<8) + (16<<16))) $((1 + (16<<8) + (16<<16))) $((4 + (16<<8) + (16<<16))) $((2 + (8<<8) + (8<<16))) $((2 + (4<<8) + (4<<16))) $((1 + (4<<8) + (8<<16))) $((2 + (8<<8) + (4<<16))); do
  echo "== tiles PX=$(printf %x $px) W=16"; PAIR_ONLY=1 MV3D_PAIR_TILES=1 MV3D_RGT_PX=$px timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair "
done
} 2>&1 | tee $OUT/tiles_ab.txt

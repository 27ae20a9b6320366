#!/bin/bash
# GPU box, round 5: one-wave tile RoiPoolGrad (divides only for hits, larger sparse map first): tile sizes, then the bench's path mode A / B
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05ai; mkdir -p $OUT
L=build_variants/libmv3d_tuning.so
{
echo "== check tiles vs plain"; MV3D_PAIR_TILES=1 NB=4 ROUNDS=2 timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "identical\|differ\|Error\|error"
echo "== old"; PAIR_ONLY=1 MV3D_PAIR_TILES=0 timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair " | tail -1
# tile pixels: fv | rgb << 8 | bev << 16 (launch order)
for px in $((2 + (16<<8) + (16<<16))) $((4 + (16<<8) + (16<<16))) $((4 + (8<<8) + (16<<16))) $((4 + (8<<8) + (8<<16))) $((2 + (8<<8) + (16<<16))) $((4 + (4<<8) + (16<<16))) $((8 + (8<<8) + (16<<16))); do
  for w in 16; do echo "== tiles PX=$(printf %x $px) W=$w"; PAIR_ONLY=1 MV3D_PAIR_TILES=1 MV3D_RGT_W=$w MV3D_RGT_PX=$px timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair " | tail -1; done
done
cp mv3d_tf_amd/libmv3d_hip.so /tmp/ship.so; cp $L mv3d_tf_amd/libmv3d_hip.so
for r in 1 2; do for t in 0 1; do
  echo "== bench path mode, MV3D_PAIR_TILES=$t run $r"
  MV3D_PAIR_TILES=$t MV3D_RGT_PX=$((4 + (8<<8) + (16<<16))) timeout 600 python bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'], d.get('verified'))"
done; done
cp /tmp/ship.so mv3d_tf_amd/libmv3d_hip.so
} 2>&1 | tee $OUT/tiles_v3b.txt

#!/bin/bash
# GPU box, round 5: the dense view as one-pixel tiles with the sum in a register (rgt_drain_pixel) against 2 x 2 LDS tiles: tests, alone, in the path
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05ba; mkdir -p $OUT
L=build_variants/libmv3d_tuning.so
PXP=$((1 + (16<<8) + (16<<16))); PXD=$((4 + (16<<8) + (16<<16)))
cp mv3d_tf_amd/libmv3d_hip.so /tmp/ship.so; cp $L mv3d_tf_amd/libmv3d_hip.so
run() { timeout 600 python bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['in_flight']['forward_us'], d['roofline']['in_flight']['backward_us'], d.get('verified',{}).get('bit_exact'))"; }
{
echo "== pair tests, dense view in pixel mode"; MV3D_RGT_PX=$PXP timeout 600 python -m pytest tests/test_roi_pair.py -x -q -m gpu 2>&1 | tail -2
echo "== all views in pixel mode"; MV3D_RGT_PX=$((1 + (1<<8) + (1<<16))) timeout 600 python -m pytest tests/test_roi_pair.py -x -q -m gpu 2>&1 | tail -2
for r in 1 2 3; do
  echo "== 2x2 LDS tiles run $r"; MV3D_RGT_PX=$PXD run
  echo "== pixel mode run $r"; MV3D_RGT_PX=$PXP run
done
} 2>&1 | tee $OUT/pixel_mode.txt
cp /tmp/ship.so mv3d_tf_amd/libmv3d_hip.so

#!/bin/bash
# GPU box, round 5: tile RoiPoolGrad with 1 / 2 / 4 channels per lane (128 / 256-channel slices per wave)
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05aj; mkdir -p $OUT
L=build_variants/libmv3d_tuning.so
{
cp mv3d_tf_amd/libmv3d_hip.so /tmp/ship.so; cp $L mv3d_tf_amd/libmv3d_hip.so
for c in 2 4; do echo "== pytest pair CPL=$c"; MV3D_PAIR_TILES=1 MV3D_RGT_CPL=$c timeout 900 python -m pytest tests/test_roi_pair.py tests/test_roipool_pin.py -x -q -m gpu 2>&1 | tail -3; done
cp /tmp/ship.so mv3d_tf_amd/libmv3d_hip.so
echo "== old"; PAIR_ONLY=1 MV3D_PAIR_TILES=0 timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair " | tail -1
PXA=$((4 + (8<<8) + (16<<16))); PXB=$((4 + (8<<8) + (8<<16))); PXC=$((2 + (16<<8) + (16<<16))); PXD=$((2 + (4<<8) + (8<<16)))
for c in 1 2 4; do for w in 8 16; do for px in $PXA $PXB $PXC $PXD; do
  echo "== tiles CPL=$c W=$w PX=$(printf %x $px)"; PAIR_ONLY=1 MV3D_PAIR_TILES=1 MV3D_RGT_CPL=$c MV3D_RGT_W=$w MV3D_RGT_PX=$px timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair \|differ\|rror" | tail -1
done; done; done
for c in 2 4; do echo "== trace CPL=$c"; MV3D_PAIR_TILES=1 MV3D_RGT_CPL=$c MV3D_RGT_PX=$PXA timeout 200 python tools/roi_tiles_trace.py --lib $L 2>&1 | grep -v amdgpu.ids; done
} 2>&1 | tee $OUT/tiles_cpl.txt

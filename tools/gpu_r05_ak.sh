#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05ak; mkdir -p $OUT
L=build_variants/libmv3d_tuning.so
{
PXA=$((4 + (8<<8) + (16<<16))); PXC=$((2 + (16<<8) + (16<<16))); PXE=$((1 + (8<<8) + (16<<16)))
echo "== check"; MV3D_PAIR_TILES=1 MV3D_RGT_CPL=1 NB=4 ROUNDS=2 timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "identical\|differ\|Error\|error"
for d in 16 0; do for c in 1 2; do for w in 16 32; do for px in $PXA $PXC $PXE; do
  echo "== tiles noprio=$d CPL=$c W=$w PX=$(printf %x $px)"; PAIR_ONLY=1 MV3D_PAIR_TILES=1 MV3D_RGT_DBG=$d MV3D_IDX_DBG=1 MV3D_RGT_CPL=$c MV3D_RGT_W=$w MV3D_RGT_PX=$px timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair \|differ\|rror" | tail -1
done; done; done; done
echo "== trace CPL=1 prio"; MV3D_PAIR_TILES=1 MV3D_RGT_CPL=1 MV3D_RGT_PX=$PXA timeout 200 python tools/roi_tiles_trace.py --lib $L 2>&1 | grep -v amdgpu.ids
} 2>&1 | tee $OUT/tiles_prio.txt

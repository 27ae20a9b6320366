#!/bin/bash
# GPU box, round 5: the forward's bin rectangles per thread in registers (LOCAL) instead of LDS + barrier -- A/B by passes
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05t; mkdir -p $OUT
L=build_variants/libmv3d_tuning.so
{
echo "== check LOCAL=1 PASSES=1 (pair vs plain outputs)"; MV3D_FWD_LOCAL=1 MV3D_FWD_PASSES=1 NB=4 ROUNDS=2 timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "identical\|differ"
echo "== check LOCAL=1 PASSES=2"; MV3D_FWD_LOCAL=1 MV3D_FWD_PASSES=2 NB=4 ROUNDS=2 timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "identical\|differ"
for r in 1 2 3; do
  echo "== LOCAL=0 run $r"; PAIR_ONLY=1 timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair "
  for ps in 1 2 4; do echo "== LOCAL=1 PASSES=$ps run $r"; PAIR_ONLY=1 MV3D_FWD_LOCAL=1 MV3D_FWD_PASSES=$ps timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair "; done
done
} | tee $OUT/fwd_local_ab.txt

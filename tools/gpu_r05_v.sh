#!/bin/bash
# GPU box, round 5: the persistent forward (next group's ROI rows requested while the current group is pooled) -- A/B by grid size
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05v; mkdir -p $OUT
L=build_variants/libmv3d_tuning.so
{
echo "== check PERSIST=2048"; MV3D_FWD_PERSIST=2048 NB=4 ROUNDS=2 timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "identical\|differ"
for r in 1 2 3; do
  for ps in 0 1024 1792 2048 3072 4096; do echo "== PERSIST=$ps run $r"; PAIR_ONLY=1 MV3D_FWD_PERSIST=$ps timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair "; done
done
} | tee $OUT/fwd_persist_ab.txt

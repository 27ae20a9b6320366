#!/bin/bash
# GPU box, round 5: the gather with whole records per workgroup (8 waves = 8 slices of ONE item), any workgroup any item
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05z; mkdir -p $OUT
L=build_variants/libmv3d_tuning.so
{
echo "== check WHOLE=1280"; MV3D_GATHER_WHOLE=1280 NB=4 ROUNDS=2 timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "identical\|differ"
for r in 1 2; do for g in 0 512 1024 1280 2048 4096; do echo "== WHOLE=$g run $r"; PAIR_ONLY=1 MV3D_GATHER_WHOLE=$g timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair "; done; done
} | tee $OUT/gather_whole_ab.txt

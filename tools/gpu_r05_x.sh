#!/bin/bash
# GPU box, round 5: the gather with two items in flight per wave -- parity (probe: pair vs plain), A/B
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05x; mkdir -p $OUT
L=build_variants/libmv3d_tuning.so
{
echo "== check GATHER2=1"; MV3D_GATHER2=1 NB=4 ROUNDS=2 timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "identical\|differ"
for r in 1 2 3; do for g in 0 1; do echo "== GATHER2=$g run $r"; PAIR_ONLY=1 MV3D_GATHER2=$g timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair "; done; done
} | tee $OUT/gather2_ab.txt

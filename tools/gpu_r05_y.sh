#!/bin/bash
# GPU box, round 5: is the gather's time its longest lists?  lists cut to 32 / 8 entries (wrong sums, timing only)
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05y; mkdir -p $OUT
L=build_variants/libmv3d_tuning.so
{
for r in 1 2; do for d in 0 32 64; do echo "== MV3D_IDX_DBG=$d run $r"; PAIR_ONLY=1 MV3D_IDX_DBG=$d timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair "; done; done
} | tee $OUT/gather_cut_lists.txt

#!/bin/bash
# (EXPERIMENTS R6.16) what the cut of the hot tiles is worth when the work list costs nothing: MV3D_RGT_PLAN=1 plans once per batch buffer (the
# list stays in the unused quarter of view 0's argmax buffer) and times the main launch alone; =2 plans in front of every call (R6.6)
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/${1:-plan}; mkdir -p $OUT
TUN=build_variants/libmv3d_tuning.so
run() { echo "-- $*"; env "$@" MV3D_IDX_DBG=1 PAIR_ONLY=1 PAIR_NO_WS=1 NB=8 ROUNDS=4 timeout 300 python tools/roi_pair_probe.py --lib $TUN 2>&1 | grep "pair \|differ\|rror" | tail -2; }
{ for r in 1 2; do
run MV3D_RGT_PLAN=0
run MV3D_RGT_PLAN=1
run MV3D_RGT_PLAN=2
for h in $PLAN_HOT; do run MV3D_RGT_PLAN=1 MV3D_RGT_HOT=$h; done
done; } 2>&1 | tee $OUT/plan.txt

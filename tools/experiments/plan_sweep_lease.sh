#!/bin/bash
# (EXPERIMENTS R6.16) the planned RoiPoolGrad alone (tuning build): cut threshold, cap, dense view's tile shape
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/${1:-plansweep}; mkdir -p $OUT
TUN=build_variants/libmv3d_tuning.so
run() { echo "-- $*"; env "$@" MV3D_IDX_DBG=1 PAIR_ONLY=1 PAIR_NO_WS=1 NB=8 ROUNDS=4 timeout 300 python tools/roi_pair_probe.py --lib $TUN 2>&1 | grep "pair \|differ\|rror" | tail -1; }
{ for r in 1 2; do
run MV3D_RGT_PLAN=1
for h in $SWEEP_HOT; do run MV3D_RGT_HOT=$h; done
for m in $SWEEP_MAX; do run MV3D_RGT_HOT_MAX=$m; done
for px in $SWEEP_PX; do run MV3D_RGT_PX=$px; done
done; } 2>&1 | tee $OUT/sweep.txt

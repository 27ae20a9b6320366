#!/bin/bash
# (EXPERIMENTS R6.16) the backward's work list written by a planning workgroup INSIDE the pair's forward launch: pair + pin + config tests on
# the production build, then timings (tuning build: MV3D_RGT_PLAN=0 static grid | 1 planned; MV3D_RGT_HOT threshold, MV3D_RGT_HOT_MAX cap)
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/${1:-planfwd}; mkdir -p $OUT
TUN=build_variants/libmv3d_tuning.so
run() { echo "-- $*"; env "$@" MV3D_IDX_DBG=1 PAIR_ONLY=1 PAIR_NO_WS=1 NB=8 ROUNDS=4 timeout 300 python tools/roi_pair_probe.py --lib $TUN 2>&1 | grep "pair \|differ\|rror" | tail -1; }
{
[ -n "${PLAN_TESTS:-1}" ] && timeout 900 python -m pytest tests/test_roi_pair.py tests/test_roipool_pin.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3
echo "-- production build"; PAIR_ONLY=1 PAIR_NO_WS=1 NB=8 ROUNDS=4 timeout 300 python tools/roi_pair_probe.py 2>&1 | grep "pair \|differ\|rror" | tail -1
for r in 1 2; do
run MV3D_RGT_PLAN=0
run MV3D_RGT_PLAN=1
for h in $PLAN_SWEEP; do run MV3D_RGT_PLAN=1 MV3D_RGT_HOT=${h%%:*} MV3D_RGT_HOT_MAX=${h##*:}; done
done; } 2>&1 | tee $OUT/planfwd.txt

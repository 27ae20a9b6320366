#!/bin/bash
# (EXPERIMENTS R6.14; source: tools/experiments/roi_grad_tiles_regacc_static_r06.hip.txt over csrc/roi_grad_tiles.hip, tools/build_tuning.sh twice:
#  plain, and MV3D_TUNING_OUT=libmv3d_regacc.so MV3D_EXTRA_FLAGS=-DRGT_REGACC_DEFAULT)
# RoiPoolGrad tiles with the sums in registers (roi_pair_tiles_kernel<W, 1, true>) against the LDS read-add-write: the pair + pin tests on a
# build that launches it by default, then timings (tuning build, MV3D_RGT_REG)
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/${1:-regacc}; mkdir -p $OUT
TUN=build_variants/libmv3d_tuning.so
run() { echo "-- $*"; env "$@" MV3D_IDX_DBG=1 PAIR_ONLY=1 PAIR_NO_WS=1 NB=8 ROUNDS=4 timeout 300 python tools/roi_pair_probe.py --lib $TUN 2>&1 | grep "pair \|differ\|rror" | tail -1; }
{
cp mv3d_tf_amd/libmv3d_hip.so /tmp/shipped.so; cp build_variants/libmv3d_regacc.so mv3d_tf_amd/libmv3d_hip.so
timeout 900 python -m pytest tests/test_roi_pair.py tests/test_roipool_pin.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3
cp /tmp/shipped.so mv3d_tf_amd/libmv3d_hip.so
for r in 1 2; do
run MV3D_RGT_REG=0
for w in $REG_W; do run MV3D_RGT_REG=1 MV3D_RGT_W=$w; done
done
for px in $REG_PX; do run MV3D_RGT_REG=0 MV3D_RGT_PX=$px; run MV3D_RGT_REG=1 MV3D_RGT_PX=$px; done
} 2>&1 | tee $OUT/regacc.txt

#!/bin/bash
# (EXPERIMENTS R6.15) RoiPoolGrad tiles with a group's bytes requested one group ahead of its adds, compiler-tracked (two named register
# sets, roi_pair_tiles_kernel<W, 1, PIPE = true>): pair + pin tests on a build that launches it by default, then timings (tuning build).
# Source: tools/experiments/roi_grad_tiles_pingpong_r06.hip.txt over csrc/roi_grad_tiles.hip, tools/build_tuning.sh twice: plain, and
# MV3D_TUNING_OUT=libmv3d_pipe8.so MV3D_EXTRA_FLAGS=-DRGT_PIPE_DEFAULT=8; PIPE_LIBS=libmv3d_pipe8.so PIPE_W="8 12 16"
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/${1:-pipe}; mkdir -p $OUT
TUN=build_variants/libmv3d_tuning.so
run() { echo "-- $*"; env "$@" MV3D_IDX_DBG=1 PAIR_ONLY=1 PAIR_NO_WS=1 NB=8 ROUNDS=4 timeout 300 python tools/roi_pair_probe.py --lib $TUN 2>&1 | grep "pair \|differ\|rror" | tail -1; }
{
for v in $PIPE_LIBS; do
cp mv3d_tf_amd/libmv3d_hip.so /tmp/shipped.so; cp build_variants/$v mv3d_tf_amd/libmv3d_hip.so
echo "== tests on $v"
timeout 900 python -m pytest tests/test_roi_pair.py tests/test_roipool_pin.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3
cp /tmp/shipped.so mv3d_tf_amd/libmv3d_hip.so
done
for r in 1 2; do
run MV3D_RGT_PIPE=0
for w in $PIPE_W; do run MV3D_RGT_PIPE=1 MV3D_RGT_W=$w; done
done
} 2>&1 | tee $OUT/pipe.txt

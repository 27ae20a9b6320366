#!/bin/bash
# RoiPoolGrad as per-pixel streams against the shipped tile kernel (EXPERIMENTS R6.13).  To reproduce: copy
# tools/experiments/roi_grad_streams_r06.hip.txt over mv3d_tf_amd/csrc/roi_grad_tiles.hip, `python -m mv3d_tf_amd.build --force` (the build then launches
# the streams kernel by default: the pair tests cover it) and tools/build_tuning.sh (MV3D_RGT_MODE / _W / _PX / _DBG), then on a lease:
#   STREAMS_W="4 8" STREAMS_DBG="1 4 2" STREAMS_PX="1052673" tools/experiments/streams_lease3.sh <tag>
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/${1:-streams}; mkdir -p $OUT
TUN=build_variants/libmv3d_tuning.so
run() { echo "-- $*"; env "$@" MV3D_IDX_DBG=1 PAIR_ONLY=1 PAIR_NO_WS=1 NB=8 ROUNDS=4 timeout 300 python tools/roi_pair_probe.py --lib $TUN 2>&1 | grep "pair \|differ\|rror" | tail -1; }
{
timeout 900 python -m pytest tests/test_roi_pair.py tests/test_roipool_pin.py -x -q -m gpu 2>&1 | tail -3
run MV3D_RGT_MODE=0
for w in $STREAMS_W; do run MV3D_RGT_MODE=1 MV3D_RGT_W=$w; done
for d in $STREAMS_DBG; do run MV3D_RGT_MODE=1 MV3D_RGT_DBG=$d; done
for px in $STREAMS_PX; do run MV3D_RGT_MODE=1 MV3D_RGT_PX=$px; done
echo "== trace"; MV3D_RGT_MODE=1 MV3D_RGT_DBG=16 timeout 300 python tools/roi_tiles_trace.py --lib $TUN 2>&1 | grep -v amdgpu.ids | tail -28
} 2>&1 | tee $OUT/streams.txt

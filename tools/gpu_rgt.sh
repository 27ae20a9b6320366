#!/bin/bash
# GPU box: tuning build of the library, parity tests of RoiPoolGrad, then tile-shape sweep with the kernel-only probe.
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/rgt; mkdir -p $OUT
MV3D_HIPCC_FLAGS=-DMV3D_TUNING python -m mv3d_tf_amd.build --force > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
if [ "${TESTS:-1}" = 1 ]; then timeout 900 python -m pytest tests/test_roipool_pin.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -15 | tee $OUT/tests.log; fi
echo "--- new default"; timeout 300 python tools/roi_bwd_probe.py 2>&1 | tail -4 | tee $OUT/new.log
for D in ${DBGS:-}; do echo "--- dbg $D"; MV3D_RT_DBG=$D ONLY=bev+rgb+fv timeout 300 python tools/roi_bwd_probe.py 2>&1 | tail -1; done | tee $OUT/dbg.log
for T in "$@"; do echo "--- tiles $T"; MV3D_BWD_TILES=$T ONLY=bev+rgb+fv timeout 300 python tools/roi_bwd_probe.py 2>&1 | tail -1 | tee -a $OUT/sweep.log; done
for Wp in ${WPGS:-}; do echo "--- wpg $Wp"; MV3D_RT_WPG=$Wp ONLY=bev+rgb+fv timeout 300 python tools/roi_bwd_probe.py 2>&1 | tail -1 | tee -a $OUT/sweep.log; done
echo "--- trace all"; timeout 300 python tools/rgt_trace.py 2>&1 | tail -25 | tee $OUT/trace.log

#!/bin/bash
# GPU box: tuning build of the library, parity checks of RoiPoolGrad, then sweeps with the kernel-only probe.
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/rgt; mkdir -p $OUT
python -m mv3d_tf_amd.build --force > /dev/null 2>&1; echo "--- product build"; timeout 300 python tools/roi_bwd_probe.py 2>&1 | tail -4 | tee $OUT/product.log
MV3D_HIPCC_FLAGS=-DMV3D_TUNING python -m mv3d_tf_amd.build --force > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout 300 python tools/rgt_debug.py 2>&1 | grep -v "^   " | tail -8
if [ "${TESTS:-1}" = 1 ]; then timeout 900 python -m pytest tests/test_roipool_pin.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -5 | tee $OUT/tests.log; fi
echo "--- new default"; timeout 300 python tools/roi_bwd_probe.py 2>&1 | tail -4 | tee $OUT/new.log
for T in "$@"; do echo "--- tiles $T"; MV3D_BWD_TILES=$T ONLY=bev+rgb+fv timeout 300 python tools/roi_bwd_probe.py 2>&1 | tail -1 | tee -a $OUT/sweep.log; done
for Wp in ${GROUPS_:-}; do echo "--- groups $Wp"; MV3D_RT_GROUPS=$Wp ONLY=bev+rgb+fv timeout 300 python tools/roi_bwd_probe.py 2>&1 | tail -1 | tee -a $OUT/sweep.log; done
for Wt in ${WEIGHTS:-}; do echo "--- weights $Wt"; MV3D_RT_WEIGHTS=$Wt ONLY=bev+rgb+fv timeout 300 python tools/roi_bwd_probe.py 2>&1 | tail -1 | tee -a $OUT/sweep.log; done
echo "--- trace all"; timeout 300 python tools/rgt_trace.py 2>&1 | tail -14 | tee $OUT/trace.log

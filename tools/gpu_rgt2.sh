#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/rgt; mkdir -p $OUT
MV3D_HIPCC_FLAGS=-DMV3D_TUNING python -m mv3d_tf_amd.build --force > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
for D in ${DBGS:-0}; do echo "--- dbg $D"; MV3D_RT_DBG=$D ONLY=${ONLYV:-bev+rgb+fv} timeout 300 python tools/roi_bwd_probe.py 2>&1 | tail -1; done | tee $OUT/dbg.log
echo "--- trace all"; timeout 300 python tools/rgt_trace.py 2>&1 | tail -25 | tee $OUT/trace.log

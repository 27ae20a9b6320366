#!/bin/bash
# GPU box, round 5: the pipelined bf16 weight-gradient kernel - parity, then A/B against the two-stage kernel (tuning build's switch)
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05o; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_conv_mfma.py -x -q -m gpu > $OUT/pytest_conv.log 2>&1; tail -5 $OUT/pytest_conv.log
L=build_variants/libmv3d_tuning.so
for r in 1 2; do for p in 0 1; do
  echo "== PIPE=$p run $r"
  MV3D_WGRAD_PIPE=$p timeout 300 python tools/wgrad_probe.py 2 --lib $L 2>&1 | tail -8
  MV3D_WGRAD_PIPE=$p timeout 300 python tools/group_probe.py bf16 2 --lib $L 2>&1 | tail -5
done; done | tee $OUT/wgrad_pipe_ab.txt
for r in 1 2; do for p in 0 1; do
  echo "== full step PIPE=$p run $r"; MV3D_WGRAD_PIPE=$p timeout 400 python tools/train_probe.py --lib $L bf16_mfma 8 2>&1 | tail -1
done; done | tee $OUT/train_step_ab.txt
tools/gpu_train_tail.sh r05o bf16_mfma 8; head -30 $OUT/bf16_mfma_tail.txt; tail -16 $OUT/bf16_mfma_tail.txt; rm -rf $OUT/tr

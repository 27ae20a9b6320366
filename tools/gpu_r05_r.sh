#!/bin/bash
# GPU box, round 5: the reduce kernel with eight loads in flight (parity + timing), multi-tensor parameter casts A/B in the bf16 step
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05r; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_conv_mfma.py tests/test_train_entry.py -x -q -m gpu > $OUT/pytest_conv.log 2>&1; tail -3 $OUT/pytest_conv.log
timeout 300 python tools/group_probe.py bf16 2 2>&1 | tail -5 | tee $OUT/group_probe.txt
for r in 1 2 3; do for c in 0 1; do
  echo "== CAST_MANY=$c run $r"; MV3D_CAST_MANY=$c timeout 400 python tools/train_probe.py bf16_mfma 12 2>&1 | tail -1
done; done | tee $OUT/cast_many_ab.txt
tools/gpu_train_tail.sh r05r bf16_mfma 8 > /dev/null 2>&1; head -30 $OUT/bf16_mfma_tail.txt | cut -c1-150; grep -A8 "^gaps" $OUT/bf16_mfma_tail.txt | cut -c1-200 | head -12; rm -rf $OUT/tr

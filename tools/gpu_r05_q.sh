#!/bin/bash
# GPU box, round 5: the graphed fusion head (unit test, clean traces of both modes), RoiPoolGrad's fill skipping item pixels (A/B)
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05q; mkdir -p $OUT
timeout 600 python -m pytest tests/test_conv_mfma.py -x -q -m gpu -k "graphed" > $OUT/pytest_head.log 2>&1; tail -3 $OUT/pytest_head.log
timeout 900 python -m pytest tests/test_roi_pair.py tests/test_gpu_configs.py -x -q -m gpu > $OUT/pytest_roi.log 2>&1; tail -3 $OUT/pytest_roi.log
L=build_variants/libmv3d_tuning.so
for r in 1 2 3; do for d in 16 0; do
  echo "== fill skip: MV3D_IDX_DBG=$d run $r"; PAIR_ONLY=1 MV3D_IDX_DBG=$d timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair "
done; done | tee $OUT/fill_skip_ab.txt
for g in 0 1; do
  MV3D_GRAPH_HEAD=$g tools/gpu_train_tail.sh r05q bf16_mfma 8 _graph$g > /dev/null 2>&1
  echo "== GRAPH_HEAD=$g"; tail -1 $OUT/bf16_mfma.log; head -34 $OUT/bf16_mfma_graph${g}_tail.txt | cut -c1-150; grep -A8 "^gaps" $OUT/bf16_mfma_graph${g}_tail.txt | cut -c1-200 | head -12
  rm -rf $OUT/tr
done

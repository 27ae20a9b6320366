#!/bin/bash
# GPU box, round 5: counters of the two bf16 weight-gradient kernels (two-stage / pipelined), the graphed fusion head: test + A/B + tail
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05p; mkdir -p $OUT
L=$GRAFT_REPO_ROOT/build_variants/libmv3d_tuning.so
for p in 0 1; do
  MV3D_WGRAD_PIPE=$p tools/gpu_r05_pmc.sh r05p/wgrad_pipe$p python $GRAFT_REPO_ROOT/tools/group_probe.py bf16 2 --lib $L > /dev/null 2>&1
  echo "== PIPE=$p"; grep -v "^#" $OUT/wgrad_pipe$p/table.txt | grep -i "wgrad_\(pipe_\)\?kernel\|^kernel " | head -12
  rm -rf $OUT/wgrad_pipe$p/*/r_results.db $OUT/wgrad_pipe$p/*/*.db
done
timeout 900 python -m pytest tests/test_conv_mfma.py tests/test_train_entry.py -x -q -m gpu -k "graphed or train_graph or determinis or train" > $OUT/pytest_head.log 2>&1; tail -5 $OUT/pytest_head.log
for r in 1 2; do for g in 0 1; do
  echo "== full step GRAPH_HEAD=$g run $r"; MV3D_GRAPH_HEAD=$g timeout 400 python tools/train_probe.py bf16_mfma 10 2>&1 | tail -1
done; done | tee $OUT/graph_head_ab.txt
echo "== fp32_mfma"; for g in 0 1; do MV3D_GRAPH_HEAD=$g timeout 400 python tools/train_probe.py fp32_mfma 6 2>&1 | tail -1; done | tee -a $OUT/graph_head_ab.txt
tools/gpu_train_tail.sh r05p bf16_mfma 8 > /dev/null 2>&1; head -12 $OUT/bf16_mfma_tail.txt | cut -c1-150; grep -A12 "^gaps" $OUT/bf16_mfma_tail.txt | cut -c1-200 | head -24; rm -rf $OUT/tr

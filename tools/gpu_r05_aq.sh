#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05aq; mkdir -p $OUT
L=build_variants/libmv3d_tuning.so
{
echo "== input layer tests"; timeout 600 python -m pytest tests/test_conv_mfma.py -x -q -m gpu -k "input_layer or conv3x3_matches or grouped" 2>&1 | tail -2
for r in 1 2; do for e in 0 1; do
  echo "== MV3D_CONV_INPUT=$e run $r"; MV3D_CONV_INPUT=$e timeout 300 python tools/conv_probe.py 16 --no-torch --lib $L --only conv1_1 2>&1 | grep conv1_1
done; done
} 2>&1 | tee $OUT/conv_input_ab2.txt

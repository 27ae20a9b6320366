#!/bin/bash
# GPU box, round 5: the input layer's own kernel (conv_input.hip) against the general kernel: tests, per layer, the serving step
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05ap; mkdir -p $OUT
L=build_variants/libmv3d_tuning.so
{
echo "== conv tests (shipped lib: input kernel on)"; timeout 1200 python -m pytest tests/test_conv_mfma.py -x -q -m gpu 2>&1 | tail -4
for r in 1 2; do for e in 0 1; do
  echo "== MV3D_CONV_INPUT=$e run $r"; MV3D_CONV_INPUT=$e timeout 300 python tools/conv_probe.py 16 --no-torch --lib $L --only conv1_1 2>&1 | grep conv1_1
done; done
for r in 1 2; do for e in 0 1; do echo "== MV3D_CONV_INPUT=$e serving step run $r"; MV3D_CONV_INPUT=$e timeout 600 python tools/serve_probe.py fp16_mfma 8 --lib $L 2>&1 | tail -1 | cut -c240-330; done; done
} 2>&1 | tee $OUT/conv_input_ab.txt

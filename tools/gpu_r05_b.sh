#!/bin/bash
# GPU box, round 5, call B: the wave-specialised 256 x 256 convolution -- numerics tests, interleaved A / B against the two-stage loop
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05b; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_conv_mfma.py -x -q -m gpu > $OUT/pytest_conv.log 2>&1; tail -4 $OUT/pytest_conv.log
for i in 1 2 3; do
  for pp in 0 1; do
    echo "== PP=$pp run $i" >> $OUT/conv_ab.txt
    MV3D_CONV_PP=$pp timeout 300 python tools/conv_probe.py 16 --no-torch --lib build_variants/libmv3d_tuning.so --only conv3_2,conv4_1,conv4_2 >> $OUT/conv_ab.txt 2>&1
  done
done
cat $OUT/conv_ab.txt | grep -v amdgpu.ids

#!/bin/bash
# GPU box, round 5: nontemporal output stores of the convolution by output size (MV3D_CONV_NT_MB): per layer and the serving step
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05ao; mkdir -p $OUT
L=build_variants/libmv3d_tuning.so
{
for mb in 100000 0 64 256 600; do
  echo "== MV3D_CONV_NT_MB=$mb layers"; MV3D_CONV_NT_MB=$mb timeout 600 python tools/conv_probe.py 16 --no-torch --lib $L 2>&1 | grep -v amdgpu.ids | grep "mfma" | awk '{printf "%s %s %s ms | ", $1, $2, $5} END {print ""}'
  for r in 1 2; do echo "== MV3D_CONV_NT_MB=$mb serving step run $r"; MV3D_CONV_NT_MB=$mb timeout 600 python tools/serve_probe.py fp16_mfma 8 --lib $L 2>&1 | tail -1 | cut -c1-400; done
done
} 2>&1 | tee $OUT/conv_nt.txt

#!/bin/bash
# GPU box, round 5: the matrix-core kernels' counters (VERDICT r04 #6a, #8) -> gpurun_out/r05l/*/table.txt
set -u
cd $GRAFT_REPO_ROOT
tools/gpu_r05_pmc.sh r05l/f16 python $GRAFT_REPO_ROOT/tools/conv_probe.py 16 --no-torch
MV3D_CONV_PP=0 tools/gpu_r05_pmc.sh r05l/f16_two_stage python $GRAFT_REPO_ROOT/tools/conv_probe.py 16 --no-torch --lib $GRAFT_REPO_ROOT/build_variants/libmv3d_tuning.so --only conv3_2,conv4_1,conv4_2
tools/gpu_r05_pmc.sh r05l/f32 python $GRAFT_REPO_ROOT/tools/conv_probe.py 16 --no-torch --f32 --only conv2_2,conv3_2,conv4_2
tools/gpu_r05_pmc.sh r05l/bf16_train python $GRAFT_REPO_ROOT/tools/group_probe.py bf16 2
tools/gpu_r05_pmc.sh r05l/f32_train python $GRAFT_REPO_ROOT/tools/group_probe.py f32 2
for d in f16 f16_two_stage f32 bf16_train f32_train; do echo "== $d"; grep -v "^#" gpurun_out/r05l/$d/table.txt | grep -i "conv3x3\|kernel " | head -40; done
# A / B: the two n-tiles of a 512-cout layer on XCD halves (MV3D_CONV_NSPLIT)
for i in 1 2 3; do for ns in 0 1; do echo "== NSPLIT=$ns run $i" >> gpurun_out/r05l/nsplit_ab.txt; MV3D_CONV_NSPLIT=$ns timeout 300 python tools/conv_probe.py 16 --no-torch --lib build_variants/libmv3d_tuning.so --only conv4_1,conv4_2 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05l/nsplit_ab.txt; done; done
cat gpurun_out/r05l/nsplit_ab.txt
